"""Ad-hoc (not a test): where the time of a cold launch_kernel-shaped call goes (800x600, one frame per scene object)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra
from raytracer_amd import abi
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "scenes")
for name in ("balls2", "dragon_large_tris", "bunny"):
    sd = ra.SceneData.load(os.path.join(G, name + ".npz"))
    for rep in range(3):
        t0 = time.perf_counter()
        sc = ra.Scene(sd)
        t1 = time.perf_counter()
        frame, st = sc.render(800, 600, 1, abi.RT_PHONG, abi.RT_JITTER_HASH)
        t2 = time.perf_counter()
        info = sc.info()
        sc.close()
        t3 = time.perf_counter()
        print(f"{name} #{rep}: create {1e3*(t1-t0):.2f} ms (host build {info['ms_build_host']:.2f}, upload {info['ms_upload']:.2f}), first frame call {1e3*(t2-t1):.2f} ms "
              f"(device {st['ms_total']:.2f}, trace {st['ms_trace']:.2f}), destroy {1e3*(t3-t2):.2f} ms", flush=True)
