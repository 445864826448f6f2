"""rt_scene_clone: a replica shares the original's host side (flattened records and trees are built once) and is an
independent scene otherwise."""
import os
import time

import numpy as np
import pytest

import raytracer_amd as ra
from raytracer_amd import abi, synth

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("name", ["recurse_simp2", "bunny", "balls2"])
def test_clone_renders_the_same_bytes(name):
    sd = ra.SceneData.load(os.path.join(GOLDEN, "scenes", name + ".npz"))
    with ra.Scene(sd) as sc:
        want, st = sc.render(320, 240, 4, abi.RT_COOK_TORRANCE, abi.RT_JITTER_HASH, seed=3)
        twin = sc.clone()
        info, info_twin = sc.info(), twin.info()
    # the original is gone; the replica lives on with the host arrays they shared
    try:
        got, st_twin = twin.render(320, 240, 4, abi.RT_COOK_TORRANCE, abi.RT_JITTER_HASH, seed=3)
        again, _ = twin.render(320, 240, 4, abi.RT_COOK_TORRANCE, abi.RT_JITTER_HASH, seed=3)
        prim, t = twin.trace_primary(320, 240, 1, abi.RT_JITTER_NONE)
        third = twin.clone()
        try:
            prim3, t3 = third.trace_primary(320, 240, 1, abi.RT_JITTER_NONE)
        finally:
            third.close()
    finally:
        twin.close()
    assert got.tobytes() == want.tobytes() and again.tobytes() == want.tobytes()
    for k in ("rays_primary", "rays_shadow", "rays_reflect", "rays_refract", "shade_nodes"):
        assert st[k] == st_twin[k], k
    for k in ("n_bvh_prims", "n_nodes", "tree_depth", "filter_depth", "device_bytes", "n_materials"):
        assert info[k] == info_twin[k], k
    assert np.array_equal(prim, prim3) and np.array_equal(t.view(np.uint32), t3.view(np.uint32))


def test_clone_keeps_its_own_camera_and_counters():
    sd = ra.SceneData.load(os.path.join(GOLDEN, "scenes", "cornell.npz"))
    with ra.Scene(sd) as sc:
        base, _ = sc.render(200, 150, 1)
        twin = sc.clone()
        try:
            cam = sd.camera
            twin.set_camera(np.asarray(cam["pos"], np.float32) + np.float32(0.5), cam["up"], cam["right"], cam["look_at"])
            moved, _ = twin.render(200, 150, 1)
            still, _ = sc.render(200, 150, 1)
        finally:
            twin.close()
    assert still.tobytes() == base.tobytes()
    assert moved.tobytes() != base.tobytes()


def test_clone_of_a_large_scene_skips_the_host_build():
    sd = synth.terrain(300, 150, lights_per_side=2)
    with ra.Scene(sd, fast_bvh=True) as sc:
        build_ms = sc.info()["ms_build_host"]
        t0 = time.perf_counter()
        twin = sc.clone()
        clone_ms = (time.perf_counter() - t0) * 1e3
        try:
            assert twin.info()["ms_build_host"] == 0.0
            a, _ = sc.render(256, 144, 1)
            b, _ = twin.render(256, 144, 1)
        finally:
            twin.close()
    assert a.tobytes() == b.tobytes()
    assert build_ms > 5.0 and clone_ms < build_ms      # (the replica pays for the upload, not for flattening and trees)


def test_host_only_scene_cannot_be_cloned():
    sd = ra.SceneData.load(os.path.join(GOLDEN, "scenes", "cornell.npz"))
    with ra.Scene(sd, host_only=True) as sc:
        with pytest.raises(Exception):
            sc.clone(0)


def test_scene_with_a_stream_of_its_own_renders_the_same_bytes(monkeypatch):
    """RT_OWN_STREAM=1 (read when a scene is created): the frame's kernels run on a stream the scene creates, bracketed by
    events on the caller's; a scene without the knob creates no stream at all."""
    sd = ra.SceneData.load(os.path.join(GOLDEN, "scenes", "recurse_simp2.npz"))
    with ra.Scene(sd) as sc:
        want, st = sc.render(320, 240, 4, abi.RT_COOK_TORRANCE, abi.RT_JITTER_HASH, seed=5)
    monkeypatch.setenv("RT_OWN_STREAM", "1")
    with ra.Scene(sd) as sc:
        got, st_own = sc.render(320, 240, 4, abi.RT_COOK_TORRANCE, abi.RT_JITTER_HASH, seed=5)
        again, _ = sc.render(320, 240, 4, abi.RT_COOK_TORRANCE, abi.RT_JITTER_HASH, seed=5)
        twin = sc.clone()
        try:
            third, _ = twin.render(320, 240, 4, abi.RT_COOK_TORRANCE, abi.RT_JITTER_HASH, seed=5)
        finally:
            twin.close()
    assert got.tobytes() == want.tobytes() and again.tobytes() == want.tobytes() and third.tobytes() == want.tobytes()
    for k in ("rays_primary", "rays_shadow", "rays_reflect", "rays_refract"):
        assert st[k] == st_own[k], k
