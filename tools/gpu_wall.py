"""Ad-hoc (not a test): wall time per frame of the headline frame as bench.py's timed loop sees it (device buffer, one frame in flight)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as ra
from raytracer_amd import abi
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "scenes")
sd = ra.SceneData.load(os.path.join(G, "bunny.npz"))
buf = torch.zeros(1920 * 1080 * 4, dtype=torch.uint8, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
p = ra.make_params(1920, 1080, 4, abi.RT_COOK_TORRANCE, abi.RT_JITTER_HASH)
with ra.Scene(sd, device=0) as sc:
    for _ in range(8):
        sc.render_device(p, buf.data_ptr(), stream)
    best = None
    for rep in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); dev = 0.0
        for _ in range(30):
            dev += sc.render_device(p, buf.data_ptr(), stream)["ms_total"]
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30 * 1e3
        best = (dt, dev / 30) if best is None or dt < best[0] else best
print("RT_SPIN_WAIT_US", os.environ.get("RT_SPIN_WAIT_US", "0"), "wall %.4f ms per frame, device %.4f" % best, flush=True)
