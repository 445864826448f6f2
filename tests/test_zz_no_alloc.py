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


def test_static_variant_allocates_no_device_memory():
    """The device's free memory (hipMemGetInfo, through torch) is the same before and after calls of the static variant.
    Several threads calling it on one device, each with a scene of its own, take turns and get the right answers."""
    import threading
    import torch
    sd = ra.SceneData.load(os.path.join(G, "cornell.npz"))
    with ra.Scene(sd) as sc:
        rays, beat = random_rays(100_000, 9)
        want_p, want_t = sc.trace_rays(rays, beat)
        sc.trace_rays(rays, beat, static=True)                  # (first use: whatever the runtime sets up lazily is set up)
        torch.cuda.synchronize()
        free0, _ = torch.cuda.mem_get_info()
        for _ in range(3):
            p, t = sc.trace_rays(rays, beat, static=True)
        torch.cuda.synchronize()
        free1, _ = torch.cuda.mem_get_info()
        assert free1 == free0
        assert np.array_equal(p, want_p) and np.array_equal(t.view(np.uint32), want_t.view(np.uint32))
    # two scenes on the one device share the array: concurrent calls must not corrupt each other
    scenes = [ra.Scene(sd), ra.Scene(sd)]
    out, errors = [None, None], []

    def work(i):
        try:
            for _ in range(4):
                out[i] = scenes[i].trace_rays(rays, beat, static=True)
        except Exception as e:      # noqa: BLE001
            errors.append(e)
    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    for s in scenes:
        s.close()
    assert not errors
    for i in range(2):
        assert np.array_equal(out[i][0], want_p) and np.array_equal(out[i][1].view(np.uint32), want_t.view(np.uint32))
