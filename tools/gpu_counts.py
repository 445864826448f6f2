"""Ad-hoc (not a test): the traversal counters of one counted frame (python tools/gpu_counts.py name [spp])."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra
from raytracer_amd import abi

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "scenes")
name = sys.argv[1]
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 1
sd = ra.SceneData.load(os.path.join(G, name + ".npz"))
with ra.Scene(sd, counters=True, miss_filter=True) as sc:
    for _ in range(2):
        _, ct = sc.render(1920, 1080, spp, abi.RT_COOK_TORRANCE if spp > 1 else abi.RT_PHONG, abi.RT_JITTER_HASH)
keys = ("rays_total", "rays_shadow", "node_visits", "prim_tests", "node_visits_shadow", "prim_tests_shadow", "wave_node_steps", "wave_leaf_steps",
        "wave_refills", "max_ray_steps", "hits_certified", "hits_contested", "hits_unreached", "shadows_certified", "tie_events_shadow", "ms_trace")
print(os.environ.get("RT_LIBRARY", "default"), name, {k: ct[k] for k in keys if k in ct}, flush=True)
