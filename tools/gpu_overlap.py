"""Ad-hoc (not a test): frames per second of the headline frame with one frame in flight and with two (two replicas of the
scene, each rendered from a thread of its own on a stream of its own: a frame's last waves and its resolve pass overlap
the next frame's first waves)."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as ra
from raytracer_amd import abi
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "scenes")
name = sys.argv[1] if len(sys.argv) > 1 else "bunny"
spp, shading = (4, ra.COOK_TORRANCE) if name.startswith("bunny") else (1, ra.PHONG)
sd = ra.SceneData.load(os.path.join(G, name + ".npz"))
p = ra.make_params(1920, 1080, spp, shading, abi.RT_JITTER_HASH)
K = 60
for n in (1, 2, 3, 1, 2):
    scenes = [ra.Scene(sd) for _ in range(n)]
    streams = [torch.cuda.Stream() for _ in range(n)]
    bufs = [torch.zeros(1920 * 1080 * 4, dtype=torch.uint8, device="cuda") for _ in range(n)]
    for sc, st, b in zip(scenes, streams, bufs):
        for _ in range(8):
            sc.render_device(p, b.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
    def work(i):
        for _ in range(K // n):
            scenes[i].render_device(p, bufs[i].data_ptr(), streams[i].cuda_stream)
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(n)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    frames = (K // n) * n
    print(f"{name}: {n} frame(s) in flight: {dt / frames * 1e3:.3f} ms per frame", flush=True)
    for sc in scenes: sc.close()
