"""Ad-hoc (not a test): ms per frame of the headline frame through rt_sequence_* with 1, 2 and 3 frames in flight."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as ra
from raytracer_amd import abi
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "scenes")
name = sys.argv[1] if len(sys.argv) > 1 else "bunny"
spp, shading = (4, ra.COOK_TORRANCE) if name.startswith("bunny") else (1, ra.PHONG)
sd = ra.SceneData.load(os.path.join(G, name + ".npz"))
p = ra.make_params(1920, 1080, spp, shading, abi.RT_JITTER_HASH)
for n in (1, 2, 3, 1, 2):
    bufs = [torch.zeros(1920 * 1080 * 4, dtype=torch.uint8, device="cuda") for _ in range(n)]
    with ra.Sequence(sd, frames_in_flight=n) as seq:
        def run(frames):
            for k in range(frames):
                if k >= n:
                    seq.wait()
                seq.submit(p, bufs[k % n].data_ptr())
            for _ in range(min(frames, n)):
                seq.wait()
        run(8 * n)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(60)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"{name}: rt_sequence with {n} frame(s) in flight: {dt / 60 * 1e3:.3f} ms per frame", flush=True)
