"""Ad-hoc: sample colours of full-size frames with the second tree (whatever RT_*_MARGIN says) against the frame without it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as ra
from raytracer_amd import abi, synth
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "scenes")
def cases():
    yield "balls2", ra.SceneData.load(os.path.join(G, "balls2.npz")), 1920, 1080, 1
    yield "iceBot", ra.SceneData.load(os.path.join(G, "iceBot.npz")), 1920, 1080, 1
    yield "cornell", ra.SceneData.load(os.path.join(G, "cornell.npz")), 1920, 1080, 1
    for seed in (78, 79, 80, 81):
        for scale in (1.0, 8.0, 60.0):
            sd = synth.random_scene(seed, n_spheres=150, n_triangles=150, n_boxes=40, n_smooth=60, n_planes=0, n_lights=2)
            sd.lights["pos"] = sd.lights["pos"] * scale
            yield f"random{seed}x{scale}", sd, 800, 600, 1
for name, sd, w, h, spp in cases():
    out = []
    for use in (True, False):
        with ra.Scene(sd, miss_filter=use) as sc:
            rgb, st = sc.render_samples(w, h, spp, abi.RT_PHONG, abi.RT_JITTER_HASH, 7)
            out.append((rgb, st))
    d = (out[0][0].view(np.uint32) != out[1][0].view(np.uint32)).any(axis=2)
    print(" ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("RT_")), name, "samples differing:", int(d.sum()), "of", d.size, "rays", out[0][1]["rays_total"], flush=True)
