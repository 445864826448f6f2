"""Ad-hoc (not a test): per-kernel device time and traversal counters of the benchmark frames on one GPU.

    python tests/gpu_breakdown.py [bunny|balls2|all]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra
from raytracer_amd import abi

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "scenes")

CASES = {
    "bunny": ("bunny", 1920, 1080, 4, abi.RT_COOK_TORRANCE),
    "bunny_obj": ("bunny_obj", 1920, 1080, 4, abi.RT_COOK_TORRANCE),
    "balls2": ("balls2", 1920, 1080, 1, abi.RT_PHONG),
    "dragon": ("dragon_large_tris", 1920, 1080, 1, abi.RT_PHONG),
}


def run(name):
    scene, w, h, spp, shading = CASES[name]
    sd = ra.SceneData.load(os.path.join(G, scene + ".npz"))
    with ra.Scene(sd) as sc:
        best = None
        for i in range(6):
            _, st = sc.render(w, h, spp, shading, abi.RT_JITTER_HASH)
            if i >= 2 and (best is None or st["ms_total"] < best["ms_total"]):
                best = st
    with ra.Scene(sd, counters=True) as sc:
        info = sc.info()
        print(f"   tree depth {info['tree_depth']}, filter depth {info['filter_depth']}", flush=True)
        for i in range(2):
            _, ct = sc.render(w, h, spp, shading, abi.RT_JITTER_HASH)
    st = best
    nv = ct["node_visits"] + ct["node_visits_shadow"]
    npt = ct["prim_tests"] + ct["prim_tests_shadow"]
    print(f"{name}: {w}x{h} spp{spp}: total {st['ms_total']:.3f} ms = trace {st['ms_trace']:.3f} + the rest {st['ms_total'] - st['ms_trace']:.3f} (counted run: light {ct['ms_light']:.3f}, combine "
          f"{ct['ms_combine']:.3f}, resolve {ct['ms_resolve']:.3f}); rays {st['rays_total'] / 1e6:.2f} M (primary {st['rays_primary'] / 1e6:.2f}, "
          f"shadow {st['rays_shadow'] / 1e6:.2f}, bounce {(st['rays_reflect'] + st['rays_refract']) / 1e6:.2f}) -> "
          f"{st['rays_total'] / st['ms_total'] / 1e3:.0f} Mrays/s", flush=True)
    print(f"   counted: trace {ct['ms_trace']:.3f} ms, node visits/ray {nv / ct['rays_total']:.1f}, prim tests/ray {npt / ct['rays_total']:.2f}, "
          f"node-step lane util {nv / max(1, 64 * ct['wave_node_steps']):.3f}, leaf-step util {npt / max(1, 64 * ct['wave_leaf_steps']):.3f}, "
          f"wave node steps {ct['wave_node_steps'] / 1e6:.2f} M, leaf steps {ct['wave_leaf_steps'] / 1e6:.2f} M, "
          f"exit mean {ct['wave_exit_mean_ms']:.3f} last {ct['wave_exit_last_ms']:.3f} ms, max list {ct['max_spawn_list']}, "
          f"attempts {ct['attempts']}", flush=True)
    slots = max(1, 64 * ct["wave_node_steps"])
    print(f"   node-step lanes: at a leaf {ct['lane_steps_at_leaf'] / slots:.3f}, without a ray {ct['lane_steps_no_ray'] / slots:.3f}; "
          f"refills {ct['wave_refills'] / 1e3:.1f} k ({ct['rays_total'] / max(1, ct['wave_refills']):.1f} rays per refill); wave time: "
          f"refill {ct['wave_time_refill']:.3f}, node steps {ct['wave_time_node']:.3f}, leaf steps {ct['wave_time_leaf']:.3f}", flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["bunny", "balls2"]
    if which == ["all"]:
        which = list(CASES)
    for n in which:
        run(n)
