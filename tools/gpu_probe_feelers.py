"""Ad-hoc (not a test): where the shadow feelers' record fetches go.  Needs a library built with
`make EXTRA=-DRT_PROBE_FEELERS` (RT_LIBRARY=...): in that build, for frames whose closest-hit rays do not use the
second tree (second_tree == 2), tie_events_shadow = records fetched by feelers in the reference's walk,
hits_contested = feelers that made that walk, hits_certified = those of them reported shadowed,
hits_unreached = all feelers reported shadowed."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra
from raytracer_amd import abi

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "scenes")
CASES = (("bunny", 4, abi.RT_COOK_TORRANCE), ("balls2", 1, abi.RT_PHONG), ("dragon_large_tris", 1, abi.RT_PHONG), ("iceBot", 1, abi.RT_PHONG))
keys = ("second_tree", "rays_shadow", "node_visits_shadow", "prim_tests_shadow", "tie_events_shadow", "hits_contested", "hits_certified",
        "hits_unreached", "shadows_certified", "ms_trace")
for name, spp, shading in CASES:
    sd = ra.SceneData.load(os.path.join(G, name + ".npz"))
    with ra.Scene(sd, counters=True) as sc:
        for i in range(5):
            _, ct = sc.render(1920, 1080, spp, shading, abi.RT_JITTER_HASH)
            if ct["second_tree"] == 2:
                print(name, {k: ct[k] for k in keys}, flush=True)
