"""rt_render's copy to the caller's frame: the first frame of a scene is copied straight into the caller's memory, later
ones through a pinned buffer of the scene's own, in parts.  The bytes are the same either way."""
import os

import numpy as np
import pytest

import raytracer_amd as ra
from raytracer_amd import abi

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_later_frames_match_the_first_at_changing_sizes():
    sd = ra.SceneData.load(os.path.join(GOLDEN, "scenes", "recurse_simp2.npz"))
    sizes = [(200, 150), (640, 480), (200, 150), (1283, 721), (64, 64), (640, 480)]      # under and over a megabyte, growing and shrinking
    first = {}
    for w, h in set(sizes):
        with ra.Scene(sd) as sc:      # a scene's first frame: the plain copy
            first[(w, h)], _ = sc.render(w, h, 1, abi.RT_PHONG, abi.RT_JITTER_NONE)
    with ra.Scene(sd) as sc:
        for w, h in sizes:
            got, _ = sc.render(w, h, 1, abi.RT_PHONG, abi.RT_JITTER_NONE)
            assert got.shape == first[(w, h)].shape and got.tobytes() == first[(w, h)].tobytes(), (w, h)


def test_pinned_copy_can_be_switched_off(monkeypatch):
    sd = ra.SceneData.load(os.path.join(GOLDEN, "scenes", "cornell.npz"))
    with ra.Scene(sd) as sc:
        want = [sc.render(640, 480, 1)[0] for _ in range(3)]
    monkeypatch.setenv("RT_PINNED_FRAME", "0")
    with ra.Scene(sd) as sc:
        got = [sc.render(640, 480, 1)[0] for _ in range(3)]
    for a, b in zip(want, got):
        assert a.tobytes() == b.tobytes() == want[0].tobytes()


def test_a_tile_part_through_the_pinned_buffer():
    sd = ra.SceneData.load(os.path.join(GOLDEN, "scenes", "cornell.npz"))
    with ra.Scene(sd) as sc:
        whole, _ = sc.render(640, 480, 1)
        parts = [sc.render(640, 480, 1, tile_first=k, tile_stride=3)[0] for k in range(3)]
    total = np.zeros_like(whole)
    for p in parts:
        assert not np.any((total != 0) & (p != 0))      # parts are disjoint; pixels outside a part read as 0
        total |= p
    assert total.tobytes() == whole.tobytes()
