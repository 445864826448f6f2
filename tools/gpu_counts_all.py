"""Ad-hoc (not a test): every statistic of one counted steady-state frame (python tools/gpu_counts_all.py name [spp] [w h])."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra
from raytracer_amd import abi, synth

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "scenes")
name = sys.argv[1]
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 1
w, h = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080)
if name == "glassy_dragon":
    sd = synth.glassy(ra.SceneData.load(os.path.join(G, "dragon_large_tris.npz")))
else:
    sd = ra.SceneData.load(os.path.join(G, name + ".npz"))
shading = abi.RT_COOK_TORRANCE if spp > 1 else abi.RT_PHONG
with ra.Scene(sd) as sc:
    for _ in range(8):
        _, st = sc.render(w, h, spp, shading, abi.RT_JITTER_HASH)
with ra.Scene(sd, counters=True) as sc:
    for _ in range(8):
        _, ct = sc.render(w, h, spp, shading, abi.RT_JITTER_HASH)
nv = ct["node_visits"] + ct["node_visits_shadow"]
npt = ct["prim_tests"] + ct["prim_tests_shadow"]
print(name, "uncounted ms_total %.3f ms_trace %.3f second_tree %d" % (st["ms_total"], st["ms_trace"], st["second_tree"]))
print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in ct.items()})
print("lane utilisation node %.3f leaf %.3f; fetches per ray %.1f; node steps per wave %.0f, leaf steps per wave %.0f, refills per wave %.0f" % (
    nv / (64.0 * max(1, ct["wave_node_steps"])), npt / (64.0 * max(1, ct["wave_leaf_steps"])), (nv + npt) / ct["rays_total"],
    ct["wave_node_steps"] / 4096.0, ct["wave_leaf_steps"] / 4096.0, ct["wave_refills"] / 4096.0), flush=True)
if ct["wave_node_steps"]:
    both, second = ct["wave_steps_both_trees"], ct["wave_steps_second_tree_only"]
    print("wave node steps with rays in both trees %.3f, in the second tree only %.3f, in the reference's only %.3f" % (
        both / ct["wave_node_steps"], second / ct["wave_node_steps"], 1.0 - (both + second) / ct["wave_node_steps"]), flush=True)
