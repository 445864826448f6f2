"""Ad-hoc (not a test; needs a library built with -DRT_DEBUG_WAVES): when each wave of the traversal launch fetched its
last primaries, spawned its last follow-up work and exited, for the last frame of a short series."""
import ctypes as C, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as ra
from raytracer_amd import abi
name = sys.argv[1] if len(sys.argv) > 1 else "balls2"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 1
sd = ra.SceneData.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "scenes", name + ".npz"))
lib = abi.load_library()
with ra.Scene(sd, miss_filter=True) as sc:
    for i in range(7):
        _, st = sc.render(1920, 1080, spp, abi.RT_COOK_TORRANCE if spp > 1 else abi.RT_PHONG, abi.RT_JITTER_HASH)
    buf = np.zeros(4096 * 16, np.uint64)
    assert lib.rt_debug_read_waves(C.c_void_p(buf.ctypes.data)) == 0
W = buf.reshape(4096, 16).astype(np.int64)
t0 = W[:, 0].min()
ms = lambda x: (x - t0) / 1e5
order = np.argsort(W[:, 2])
print("trace", st["ms_trace"], "ms; exits: first", round(ms(W[order[0], 2]), 2), "median", round(ms(W[order[2048], 2]), 2), "p90", round(ms(W[order[3686], 2]), 2),
      "p99", round(ms(W[order[4055], 2]), 2), "last", round(ms(W[order[-1], 2]), 2))
print("last primaries fetched: median", round(float(np.median(ms(W[:, 1]))), 2), "max", round(float(ms(W[:, 1]).max()), 2))
KIND = {0: "closest-hit", 1: "feeler", 2: "feeler(set-up)"}
def last_ray(w):
    mode, tag, born = int(w[10]) & 0xff, (int(w[10]) >> 8) & 0xff, int(w[10]) >> 16
    return (f"{KIND[mode & 3]}{' in second tree' if mode & 4 else ''}{' verifying' if mode & 8 else ''} tag {tag}, cost {born}, "
            f"started {ms(w[9]):.2f}, took {(w[8] - w[9]) / 1e5:.3f} ms")
print("wave, exit ms, last fetch ms, last spawn ms, list peak, list when dry, slots fetched, refills | the ray it finished last")
for k in order[-40:]:
    w = W[k]
    print(k, round(ms(w[2]), 2), round(ms(w[1]), 2), round(ms(w[6]), 2), w[3], w[7], w[4], w[5], "|", last_ray(w))
took = (W[:, 8] - W[:, 9]) / 1e5
cost = W[:, 10] >> 16
print("last rays of all waves: took median", round(float(np.median(took)), 3), "p90", round(float(np.percentile(took, 90)), 3), "max", round(float(took.max()), 3),
      "ms; cost median", int(np.median(cost)), "p90", int(np.percentile(cost, 90)), "max", int(cost.max()),
      "; feelers", int(((W[:, 10] & 3) != 0).sum()), "of", len(W))
late = order[-400:]
print("last rays of the last 400 waves: took median", round(float(np.median(took[late])), 3), "cost median", int(np.median(cost[late])),
      "feelers", int(((W[late, 10] & 3) != 0).sum()), "; ms per cost unit", round(float(np.median(took[late] / np.maximum(cost[late], 1))) * 1e3, 2), "us")
print("medians: list peak", np.median(W[:, 3]), "list when dry", np.median(W[:, 7]), "fetched", np.median(W[:, 4]))
