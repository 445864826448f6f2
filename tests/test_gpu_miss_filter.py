"""The miss filter changes no result.

Beside the reference's tree a scene gets a second, widened surface-area tree that the traversal kernel uses only to
prove that a ray meets no primitive (closest-hit rays) or none in front of its target (shadow feelers); every other
ray walks the reference's tree as before.  So frames, primary hits and arbitrary-ray hits must be the same bits with
and without it -- on scenes of every primitive class, with planes (which the filter's feeler rule treats apart), with
transforms, and on small scenes that only get a filter because the test lowers the threshold.
"""
import os

import numpy as np
import pytest

import raytracer_amd as ra
from raytracer_amd import abi, synth

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scenes")


@pytest.fixture(autouse=True)
def small_scenes_get_a_filter():
    old = os.environ.get("RT_FILTER_MIN_PRIMS")
    os.environ["RT_FILTER_MIN_PRIMS"] = "2"
    yield
    if old is None:
        os.environ.pop("RT_FILTER_MIN_PRIMS", None)
    else:
        os.environ["RT_FILTER_MIN_PRIMS"] = old


def both_ways(sd, w, h, spp, shading, jitter=abi.RT_JITTER_HASH, seed=5):
    out = []
    for use in (True, False):
        with ra.Scene(sd, miss_filter=use) as sc:
            info = sc.info()
            assert (info["filter_depth"] > 0) == (use and info["n_bvh_prims"] >= 2)
            frame, st = sc.render(w, h, spp, shading, jitter, seed)
            prim, t = sc.trace_primary(w, h, spp, jitter, seed)
            rgb, _ = sc.render_samples(w, h, spp, shading, jitter, seed)
            out.append((frame, prim, t, rgb, st))
    (f1, p1, t1, c1, s1), (f0, p0, t0, c0, s0) = out
    assert np.array_equal(p1, p0)
    assert np.array_equal(t1.view(np.uint32), t0.view(np.uint32))
    assert np.array_equal(c1.view(np.uint32), c0.view(np.uint32))
    assert f1.tobytes() == f0.tobytes()
    for k in ("rays_primary", "rays_shadow", "rays_reflect", "rays_refract", "shade_nodes", "shadow_skipped"):
        assert s1[k] == s0[k], k
    return s1


@pytest.mark.parametrize("name", ["cornell", "balls2", "bunny", "iceBot", "dragon_large_tris", "box", "transforms", "final2", "recurse_simp2",
                                  "simple_refract", "valentine", "gnarly", "shark", "miniShark", "monkey", "blitz", "sphere_bunny_small", "recurses", "donut"])
@pytest.mark.parametrize("shading", [abi.RT_PHONG, abi.RT_COOK_TORRANCE])
def test_frames_equal_with_and_without_filter(name, shading):
    sd = ra.SceneData.load(os.path.join(G, name + ".npz"))
    both_ways(sd, 320, 240, 1, shading)


@pytest.mark.parametrize("seed", range(8))
def test_random_scenes_equal_with_and_without_filter(seed):
    both_ways(synth.random_scene(seed, n_planes=2), 200, 150, 4, abi.RT_PHONG)


def test_many_lights_and_glass_equal_with_and_without_filter():
    both_ways(synth.terrain(100, 50), 160, 90, 1, abi.RT_PHONG)
    both_ways(synth.glassy(ra.SceneData.load(os.path.join(G, "dragon_large_tris.npz")), everything=True), 200, 120, 1, abi.RT_PHONG)


@pytest.mark.parametrize("name", ["cornell", "balls2", "iceBot"])
def test_arbitrary_rays_equal_with_and_without_filter(name):
    sd = ra.SceneData.load(os.path.join(G, name + ".npz"))
    rng = np.random.Generator(np.random.PCG64(11))
    n = 100_000
    o = (rng.random((n, 3)) - 0.5) * 14.0
    d = rng.normal(size=(n, 3))
    d[::13, 1] = 0.0
    rays = np.concatenate([o, d], axis=1).astype(np.float32)
    beat = np.where(rng.random(n) < 0.5, -3.4028234663852886e38, rng.random(n) * 6.0).astype(np.float32)
    got = []
    for use in (True, False):
        with ra.Scene(sd, miss_filter=use) as sc:
            got.append(sc.trace_rays(rays, beat))
    assert np.array_equal(got[0][0], got[1][0])
    assert np.array_equal(got[0][1].view(np.uint32), got[1][1].view(np.uint32))
