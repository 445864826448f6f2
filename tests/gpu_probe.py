"""Ad-hoc GPU probe (not a test): timings and traversal statistics for a scene."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as ra
from raytracer_amd import abi
name = sys.argv[1] if len(sys.argv) > 1 else "balls2"
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
shading = int(sys.argv[4]) if len(sys.argv) > 4 else 0
samples = int(sys.argv[5]) if len(sys.argv) > 5 else 1
sd = ra.SceneData.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scenes", name + ".npz"))
with ra.Scene(sd) as sc:
    for i in range(3):
        frame, st = sc.render(w, h, samples, shading, abi.RT_JITTER_HASH)
    print(name, w, h, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items() if not k.startswith(("node", "prim", "tie", "wave"))})
    print(f"  Mrays/s {st['rays_total'] / st['ms_total'] / 1e3:.1f}")
with ra.Scene(sd, counters=True) as sc:
    frame, st = sc.render(w, h, samples, shading, abi.RT_JITTER_HASH)
    n_ext = st['rays_primary'] + st['rays_reflect'] + st['rays_refract']
    print(f"  extend: visits/ray {st['node_visits']/max(1,n_ext):.1f} prim/ray {st['prim_tests']/max(1,n_ext):.2f} lane-util node {st['node_visits']/max(1,64*st['wave_node_steps']):.3f} leaf {st['prim_tests']/max(1,64*st['wave_leaf_steps']):.3f} ties {st['tie_events']} max-ray-steps {st['max_ray_steps']} wave-steps {st['wave_node_steps']+st['wave_leaf_steps']}")
    print(f"  shadow: visits/ray {st['node_visits_shadow']/max(1,st['rays_shadow']):.1f} prim/ray {st['prim_tests_shadow']/max(1,st['rays_shadow']):.2f} lane-util node {st['node_visits_shadow']/max(1,64*st['wave_node_steps_shadow']):.3f} leaf {st['prim_tests_shadow']/max(1,64*st['wave_leaf_steps_shadow']):.3f} max-ray-steps {st['max_ray_steps_shadow']} wave-steps {st['wave_node_steps_shadow']+st['wave_leaf_steps_shadow']}")
