"""rt_trace_rays_static -- the variant of rt_trace_rays that allocates no device memory -- against rt_trace_rays:
same primitive ids and the same bits of t, on two calls in a row through the C ABI, the second with fewer rays
(the static working memory keeps the first call's contents), plus the size limit."""
import os

import numpy as np
import pytest

import raytracer_amd as ra
from raytracer_amd import abi

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scenes")


def random_rays(n, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    o = (rng.random((n, 3)) - 0.5) * 12.0
    d = rng.normal(size=(n, 3))
    d[::17, 0] = 0.0                      # some axis-parallel directions
    to_beat = np.where(rng.random(n) < 0.5, -3.4028234663852886e38, rng.random(n) * 8.0)
    return np.concatenate([o, d], axis=1).astype(np.float32), to_beat.astype(np.float32)


@pytest.mark.parametrize("name", ["cornell", "balls2"])
def test_static_variant_equals_allocating_one(name):
    sd = ra.SceneData.load(os.path.join(G, name + ".npz"))
    with ra.Scene(sd) as sc:
        for n, seed, use_beat in ((200_000, 1, True), (5_003, 2, False), (64, 3, True)):
            rays, beat = random_rays(n, seed)
            tb = beat if use_beat else None
            p0, t0 = sc.trace_rays(rays, tb)
            p1, t1 = sc.trace_rays(rays, tb, static=True)
            assert np.array_equal(p0, p1)
            assert np.array_equal(t0.view(np.uint32), t1.view(np.uint32))
            assert (p0 >= 0).any() and (p0 < 0).any()


def test_static_variant_rejects_too_many_rays():
    sd = ra.SceneData.load(os.path.join(G, "simple.npz"))
    with ra.Scene(sd) as sc:
        rays = np.zeros((abi.RT_TRACE_RAYS_STATIC_MAX + 1, 6), np.float32)
        rays[:, 5] = 1.0
        with pytest.raises(ra.RaytracerError) as e:
            sc.trace_rays(rays, static=True)
        assert e.value.code == abi.RT_ERR_TOO_LARGE
        p, t = sc.trace_rays(rays[:abi.RT_TRACE_RAYS_STATIC_MAX], static=True)       # the limit itself is accepted
        assert len(p) == abi.RT_TRACE_RAYS_STATIC_MAX
