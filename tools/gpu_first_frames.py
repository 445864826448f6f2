"""Ad-hoc (not a test): the first frames of a fresh scene object against its steady state (python tools/gpu_first_frames.py name [w h])."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra
from raytracer_amd import abi
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "scenes")
name = sys.argv[1] if len(sys.argv) > 1 else "balls2"
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (800, 600)
sd = ra.SceneData.load(os.path.join(G, name + ".npz"))
for rep in range(3):
    with ra.Scene(sd) as sc:
        out = []
        for i in range(10):
            _, st = sc.render(w, h, 1, abi.RT_PHONG, abi.RT_JITTER_HASH)
            out.append("%.2f/%.2f(%d,%d)" % (st["ms_total"], st["ms_trace"], st["second_tree"], st["attempts"]))
    print(name, w, h, "ms_total/ms_trace(second_tree,attempts) per frame:", " ".join(out), flush=True)
