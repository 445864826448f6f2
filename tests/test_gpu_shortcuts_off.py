"""Every shortcut on against every shortcut off, on random scenes seen from random cameras.

With the defaults a frame uses the second tree's proofs, leaves out tiles nothing can be seen through, stops feelers
whose answer cannot matter after their own-surface test and starts its tiles costliest first.  With RT_MISS_FILTER=0
RT_CULL_TILES=0 RT_SKIP_UNLIT=0 RT_TILE_ORDER=0 every ray makes the reference's walk of the reference's tree
(kernel.cu:83-200) in image order.  Frames must be the same bytes and the ray counts equal.
(tools/gpu_stress.py is the same comparison over as many seeds as one likes.)
"""
import os

import numpy as np
import pytest

import raytracer_amd as ra
from raytracer_amd import abi, synth

pytestmark = pytest.mark.gpu
OFF = {"RT_MISS_FILTER": "0", "RT_CULL_TILES": "0", "RT_SKIP_UNLIT": "0", "RT_TILE_ORDER": "0"}
COUNTS = ("rays_total", "rays_primary", "rays_shadow", "rays_reflect", "rays_refract", "shade_nodes", "shadow_skipped")


def render(sd, env, w, h, spp, shading, frames):
    env = dict(env, RT_FILTER_MIN_PRIMS="4")          # small scenes get a second tree too
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        with ra.Scene(sd) as sc:
            for _ in range(frames):
                f, st = sc.render(w, h, spp, shading, abi.RT_JITTER_HASH, 3)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return f, st


@pytest.mark.parametrize("block", range(6))
def test_random_scenes_and_views(block):
    used = 0
    for seed in range(5000 + 10 * block, 5010 + 10 * block):
        rng = np.random.Generator(np.random.PCG64(10_000 + seed))
        sd = synth.random_scene(seed, n_spheres=int(rng.integers(0, 40)), n_triangles=int(rng.integers(0, 60)), n_boxes=int(rng.integers(0, 12)),
                                n_smooth=int(rng.integers(0, 40)), n_planes=int(rng.integers(0, 2)), n_lights=int(rng.integers(1, 6)))
        if len(sd.spheres) + len(sd.triangles) + len(sd.boxes) + len(sd.smooth_triangles) == 0:
            continue
        pos = rng.normal(size=3) * rng.choice([1.0, 6.0, 20.0])
        at = rng.normal(size=3) * rng.choice([0.5, 3.0])
        sd.camera = {"pos": pos.astype(np.float32), "up": np.array([0, 1, 0], np.float32),
                     "right": np.array([rng.choice([1.0, 1.3333, 1.7778]), 0, 0], np.float32), "look_at": at.astype(np.float32)}
        w, h = int(rng.integers(40, 400)), int(rng.integers(40, 300))
        spp = int(rng.choice([1, 1, 4, 9]))
        shading = abi.RT_PHONG if rng.random() < 0.6 else abi.RT_COOK_TORRANCE
        frames = int(rng.choice([1, 2, 3]))
        f1, s1 = render(sd, {}, w, h, spp, shading, frames)
        f0, s0 = render(sd, OFF, w, h, spp, shading, 1)
        assert f1.tobytes() == f0.tobytes(), (seed, w, h, spp, shading, frames)
        for k in COUNTS:
            assert s1[k] == s0[k], (seed, k)
        assert s0["primaries_culled"] == 0 and s0["shadows_unneeded"] == 0 and s0["hits_certified"] == 0
        used += 1
    assert used >= 8
