"""Ad-hoc (not a test): node visits by ray kind, with and without the miss filter."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra
from raytracer_amd import abi
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "scenes")
for name, spp, sh in (("balls2", 1, abi.RT_PHONG), ("bunny", 4, abi.RT_COOK_TORRANCE), ("dragon_large_tris", 1, abi.RT_PHONG), ("iceBot", 1, abi.RT_PHONG)):
    sd = ra.SceneData.load(os.path.join(G, name + ".npz"))
    for filt in (True, False):
        with ra.Scene(sd, counters=True, miss_filter=filt) as sc:
            _, ct = sc.render(1920, 1080, spp, sh, abi.RT_JITTER_HASH)
        ne = ct["rays_primary"] + ct["rays_reflect"] + ct["rays_refract"]
        ns = ct["rays_shadow"]
        print(f"{name} filter={filt}: extend rays {ne/1e6:.2f} M visits/ray {ct['node_visits']/max(1,ne):.1f} tests/ray {ct['prim_tests']/max(1,ne):.2f} ties {ct['tie_events']}; "
              f"feelers {ns/1e6:.2f} M visits/ray {ct['node_visits_shadow']/max(1,ns):.1f} tests/ray {ct['prim_tests_shadow']/max(1,ns):.2f}; shade nodes {ct['shade_nodes']/1e6:.2f} M; trace {ct['ms_trace']:.2f} ms; "
              f"hits certified {ct['hits_certified']/1e6:.2f} M, with a rival {ct['hits_contested']/1e6:.2f} M, unreached {ct['hits_unreached']/1e6:.2f} M", flush=True)
