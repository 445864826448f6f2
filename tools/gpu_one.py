"""Ad-hoc (not a test): steady-state frame times of chosen benchmark frames (python tools/gpu_one.py name[,name...] [frames])."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra
from raytracer_amd import abi

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "scenes")
SPP = {"bunny": (4, abi.RT_COOK_TORRANCE), "bunny_obj": (4, abi.RT_COOK_TORRANCE)}
names = sys.argv[1].split(",")
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 14
out = []
for name in names:
    spp, shading = SPP.get(name, (1, abi.RT_PHONG))
    sd = ra.SceneData.load(os.path.join(G, name + ".npz"))
    with ra.Scene(sd) as sc:
        ts = []
        for i in range(frames):
            _, st = sc.render(1920, 1080, spp, shading, abi.RT_JITTER_HASH)
            if i >= 5:
                ts.append((st["ms_total"], st["ms_trace"]))
    ts.sort()
    out.append(f"{name} {ts[0][0]:.3f}/{ts[len(ts) // 2][0]:.3f} (trace {ts[0][1]:.3f})")
print(" ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("RT_")), "| min/median ms:", "; ".join(out), flush=True)
