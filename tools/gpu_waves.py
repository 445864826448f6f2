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
        _, st = sc.render(1920, 1080, spp, abi.RT_PHONG, abi.RT_JITTER_HASH)
    buf = np.zeros(4096 * 8, np.uint64)
    assert lib.rt_debug_read_waves(C.c_void_p(buf.ctypes.data)) == 0
W = buf.reshape(4096, 8).astype(np.int64)
t0 = W[:, 0].min()
ms = lambda x: (x - t0) / 1e5
order = np.argsort(W[:, 2])
print("trace", st["ms_trace"], "ms; exits: first", round(ms(W[order[0], 2]), 2), "median", round(ms(W[order[2048], 2]), 2), "p90", round(ms(W[order[3686], 2]), 2),
      "p99", round(ms(W[order[4055], 2]), 2), "last", round(ms(W[order[-1], 2]), 2))
print("last primaries fetched: median", round(float(np.median(ms(W[:, 1]))), 2), "max", round(float(ms(W[:, 1]).max()), 2))
print("wave, exit ms, last fetch ms, last spawn ms, list peak, list when dry, slots fetched, refills")
for k in order[-30:]:
    w = W[k]
    print(k, round(ms(w[2]), 2), round(ms(w[1]), 2), round(ms(w[6]), 2), w[3], w[7], w[4], w[5])
print("medians: list peak", np.median(W[:, 3]), "list when dry", np.median(W[:, 7]), "fetched", np.median(W[:, 4]))
