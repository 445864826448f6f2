"""Ad-hoc (not a test): steady-state frame times of the benchmark frames, for A/B runs of two builds of the library
(RT_LIBRARY=... python tests/gpu_ab.py [frames])."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra
from raytracer_amd import abi

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "scenes")
CASES = (("bunny", 4, abi.RT_COOK_TORRANCE), ("bunny_obj", 4, abi.RT_COOK_TORRANCE), ("balls2", 1, abi.RT_PHONG),
         ("dragon_large_tris", 1, abi.RT_PHONG), ("iceBot", 1, abi.RT_PHONG))
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 14
out = []
for name, spp, shading in CASES:
    sd = ra.SceneData.load(os.path.join(G, name + ".npz"))
    with ra.Scene(sd) as sc:
        ts = []
        for i in range(frames):
            _, st = sc.render(1920, 1080, spp, shading, abi.RT_JITTER_HASH)
            if i >= 5:
                ts.append((st["ms_total"], st["ms_trace"]))
    choice = st["second_tree"]
    ts.sort()
    out.append(f"{name} {ts[0][0]:.3f}/{ts[len(ts) // 2][0]:.3f} (trace {ts[0][1]:.3f}, second tree {choice})")
print(os.environ.get("RT_LIBRARY", "default"), "| min/median ms:", "; ".join(out), flush=True)
