"""Ad-hoc (not a test): the trace time of every frame of a run, to see how the tile start order settles."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra
from raytracer_amd import abi
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "scenes")
name = sys.argv[1] if len(sys.argv) > 1 else "balls2"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 1
sd = ra.SceneData.load(os.path.join(G, name + ".npz"))
with ra.Scene(sd) as sc:
    ts = []
    for i in range(16):
        _, st = sc.render(1920, 1080, spp, abi.RT_PHONG, abi.RT_JITTER_HASH)
        ts.append(st["ms_trace"])
print(os.environ.get("RT_LIBRARY", "default")[-8:], os.environ.get("RT_TILE_ORDER", "1"), " ".join(f"{t:.2f}" for t in ts), flush=True)
