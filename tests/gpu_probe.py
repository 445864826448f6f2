"""Ad-hoc GPU probe (not a test): timings and traversal statistics for a scene.

    python tests/gpu_probe.py [scene [width height [shading [samples]]]]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra
from raytracer_amd import abi

name = sys.argv[1] if len(sys.argv) > 1 else "balls2"
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
shading = int(sys.argv[4]) if len(sys.argv) > 4 else 0
samples = int(sys.argv[5]) if len(sys.argv) > 5 else 1
fast = os.environ.get("PROBE_FAST_BVH", "0") == "1"
gpu = os.environ.get("PROBE_GPU_BVH", "0") == "1"
sd = ra.SceneData.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scenes", name + ".npz"))
with ra.Scene(sd, fast_bvh=fast, gpu_bvh=gpu) as sc:
    print(sc.info())
    best = None
    for i in range(5):
        frame, st = sc.render_device_host(w, h, samples, shading, abi.RT_JITTER_HASH) if hasattr(sc, "render_device_host") else sc.render(w, h, samples, shading, abi.RT_JITTER_HASH)
        if i and (best is None or st["ms_total"] < best["ms_total"]):
            best = st
    st = best
    print(f"{name} {w}x{h} spp{samples}: ms_total {st['ms_total']:.3f}  trace {st['ms_trace']:.3f} ({st['launches_trace']} launches)  "
          f"light {st['ms_light']:.3f} combine {st['ms_combine']:.3f} resolve {st['ms_resolve']:.3f}  "
          f"rays {st['rays_total']} -> {st['rays_total'] / st['ms_total'] / 1e3:.1f} Mrays/s")
with ra.Scene(sd, counters=True, fast_bvh=fast, gpu_bvh=gpu) as sc:
    frame, st = sc.render(w, h, samples, shading, abi.RT_JITTER_HASH)
    n_ext = st['rays_primary'] + st['rays_reflect'] + st['rays_refract']
    nv, npr = st['node_visits'] + st['node_visits_shadow'], st['prim_tests'] + st['prim_tests_shadow']
    print(f"  extend: visits/ray {st['node_visits'] / max(1, n_ext):.1f} prim/ray {st['prim_tests'] / max(1, n_ext):.2f} ties {st['tie_events']}   "
          f"shadow: visits/ray {st['node_visits_shadow'] / max(1, st['rays_shadow']):.1f} prim/ray {st['prim_tests_shadow'] / max(1, st['rays_shadow']):.2f}")
    print(f"  wave steps: node {st['wave_node_steps']} leaf {st['wave_leaf_steps']}  lane-util node {nv / max(1, 64 * st['wave_node_steps']):.3f} "
          f"leaf {(npr + st['rays_shadow'] + st['shadow_skipped']) / max(1, 64 * st['wave_leaf_steps']):.3f}  max-ray-steps {st['max_ray_steps']}")
    print(f"  counted launch {st['ms_trace']:.3f} ms: waves done at mean {st['wave_exit_mean_ms']:.3f} ms, last {st['wave_exit_last_ms']:.3f} ms")
