"""Ad-hoc (not a test): the glassy dragon at 3840x2160, depth 6 -- frame time and traversal counters."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra
from raytracer_amd import abi, synth
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "scenes")
sd = synth.glassy(ra.SceneData.load(os.path.join(G, "dragon_large_tris.npz")))
with ra.Scene(sd) as sc:
    for i in range(8):
        _, st = sc.render(3840, 2160, 1, abi.RT_PHONG, abi.RT_JITTER_HASH)
    print("uncounted", {k: st[k] for k in ("ms_total", "ms_trace", "rays_primary", "rays_shadow", "rays_reflect", "rays_refract", "shade_nodes", "second_tree", "primaries_culled", "shadows_unneeded")})
with ra.Scene(sd, counters=True) as sc:
    for i in range(7):
        _, ct = sc.render(3840, 2160, 1, abi.RT_PHONG, abi.RT_JITTER_HASH)
    nv = ct["node_visits"] + ct["node_visits_shadow"]; npt = ct["prim_tests"] + ct["prim_tests_shadow"]
    print("counted", {k: ct[k] for k in ("ms_trace", "ms_light", "ms_combine", "ms_resolve", "node_visits", "node_visits_shadow", "prim_tests", "prim_tests_shadow", "wave_node_steps", "wave_leaf_steps", "wave_refills",
          "wave_exit_mean_ms", "wave_exit_last_ms", "hits_certified", "hits_contested", "hits_unreached", "shadows_certified", "max_ray_steps", "max_spawn_list", "wave_time_refill", "wave_time_node", "wave_time_leaf", "lane_steps_at_leaf", "lane_steps_no_ray", "second_tree")})
    print("util node", nv / (64 * ct["wave_node_steps"]), "leaf", npt / (64 * ct["wave_leaf_steps"]), "visits/ray", nv / ct["rays_total"])
