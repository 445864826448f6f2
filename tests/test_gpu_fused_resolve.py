"""Shading in the resolve pass changes no byte.

Tiles none of whose primaries spawned a bounce ray skip the light-sum and combine kernels; the resolve kernel shades
their samples itself with the same functions.  Frames must be the same bytes either way (RT_FUSE_RESOLVE=0 turns the
shortcut off, read when a scene is created) -- with bounce rays in some tiles only, in none, in all; with frame edges
that cut tiles; with anti-aliasing; in several chunks; in tile parts -- and the frame must still be the average of the
per-sample colours the unfused path reports.
"""
import os

import numpy as np
import pytest

import raytracer_amd as ra
from raytracer_amd import abi, synth

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scenes")


def scene_with(fuse, sd, **kw):
    old = os.environ.get("RT_FUSE_RESOLVE")
    os.environ["RT_FUSE_RESOLVE"] = "1" if fuse else "0"
    try:
        return ra.Scene(sd, **kw)
    finally:
        if old is None:
            os.environ.pop("RT_FUSE_RESOLVE", None)
        else:
            os.environ["RT_FUSE_RESOLVE"] = old


def frames(sd, w, h, spp, shading, n=1, **kw):
    out = []
    for fuse in (True, False):
        with scene_with(fuse, sd, **kw) as sc:
            last = None
            for _ in range(n):
                last, st = sc.render(w, h, spp, shading, abi.RT_JITTER_HASH, 3)
            out.append((last, st))
    return out


@pytest.mark.parametrize("name", ["cornell", "balls2", "bunny", "iceBot", "dragon_large_tris", "recurse_simp2", "simple_refract", "valentine", "blitzcrank"])
@pytest.mark.parametrize("shading", [abi.RT_PHONG, abi.RT_COOK_TORRANCE])
def test_same_bytes_fused_and_unfused(name, shading):
    path = os.path.join(G, name + ".npz")
    if not os.path.exists(path):
        pytest.skip("no such fixture scene")
    sd = ra.SceneData.load(path)
    (f1, s1), (f0, s0) = frames(sd, 333, 257, 1, shading)      # edges cut the last tile column and row
    assert f1.tobytes() == f0.tobytes()
    assert s1["rays_total"] == s0["rays_total"]


def test_same_bytes_with_anti_aliasing_and_a_second_frame():
    """The second frame uses the tile order made from the first one's costs, which the resolve pass now helps to record."""
    sd = ra.SceneData.load(os.path.join(G, "balls2.npz"))
    (f1, _), (f0, _) = frames(sd, 400, 300, 4, abi.RT_PHONG, n=3)
    assert f1.tobytes() == f0.tobytes()


def test_same_bytes_in_chunks_and_parts():
    sd = ra.SceneData.load(os.path.join(G, "balls2.npz"))
    w, h = 320, 256
    with scene_with(False, sd) as sc:
        ref, _ = sc.render(w, h, 1, abi.RT_PHONG, abi.RT_JITTER_HASH, 3)
    old = os.environ.get("RT_CHUNK_TILES")
    os.environ["RT_CHUNK_TILES"] = "7"
    try:
        with scene_with(True, sd) as sc:
            got, st = sc.render(w, h, 1, abi.RT_PHONG, abi.RT_JITTER_HASH, 3)
            assert st["launches_trace"] > 1
    finally:
        if old is None:
            os.environ.pop("RT_CHUNK_TILES", None)
        else:
            os.environ["RT_CHUNK_TILES"] = old
    assert got.tobytes() == ref.tobytes()


def test_many_lights_without_bounces():
    sd = synth.terrain(60, 40, lights_per_side=6)
    (f1, _), (f0, _) = frames(sd, 256, 160, 4, abi.RT_PHONG, fast_bvh=True)
    assert f1.tobytes() == f0.tobytes()


def test_frame_is_the_average_of_the_samples():
    sd = ra.SceneData.load(os.path.join(G, "bunny.npz"))
    w, h, spp = 200, 150, 4
    with scene_with(True, sd) as sc:
        frame, _ = sc.render(w, h, spp, abi.RT_COOK_TORRANCE, abi.RT_JITTER_HASH, 1)
        rgb, _ = sc.render_samples(w, h, spp, abi.RT_COOK_TORRANCE, abi.RT_JITTER_HASH, 1)
    s = int(round(spp ** 0.5))
    acc = np.zeros((h, w, 3), np.float32)
    for xo in range(s):                 # the order of averageBufferColors, kernel.cu:556-560
        for yo in range(s):
            acc = acc + rgb[yo::s, xo::s, :]
    px = (acc / np.float32(spp) * np.float32(255.0))
    want = np.clip(np.nan_to_num(px, nan=0.0).astype(np.int64), 0, 255).astype(np.uint8)
    assert np.array_equal(frame[:, :, :3], want)


@pytest.mark.parametrize("fuse", [True, False])
@pytest.mark.parametrize("name", ["balls2", "bunny"])
def test_nothing_is_read_from_the_frame_before(name, fuse):
    """A primary that hits nothing stores only its hit; its spawn record and cost keep whatever an earlier frame (another
    camera, another frame size) left in the queue memory, and nothing may read them."""
    sd = ra.SceneData.load(os.path.join(G, name + ".npz"))
    cam = sd.camera
    cam2 = {"pos": cam["pos"] * np.float32(1.6) + np.array([0.4, 0.2, -0.3], np.float32), "up": cam["up"], "right": cam["right"],
            "look_at": cam["look_at"]}
    sd2 = ra.SceneData.load(os.path.join(G, name + ".npz"))
    sd2.camera = cam2
    with scene_with(fuse, sd2) as sc:
        want, _ = sc.render(300, 200, 1, abi.RT_PHONG, abi.RT_JITTER_HASH, 5)
    with scene_with(fuse, sd) as sc:
        sc.render(300, 200, 1, abi.RT_PHONG, abi.RT_JITTER_HASH, 5)
        sc.render(420, 330, 4, abi.RT_PHONG, abi.RT_JITTER_HASH, 5)         # other slots, other hits
        sc.set_camera(cam2["pos"], cam2["up"], cam2["right"], cam2["look_at"])
        got, _ = sc.render(300, 200, 1, abi.RT_PHONG, abi.RT_JITTER_HASH, 5)
    assert got.tobytes() == want.tobytes()


def test_tile_subset_after_a_whole_frame():
    """rt_render keeps its device copy of the frame from call to call: pixels of tiles a later call does not select are 0."""
    sd = ra.SceneData.load(os.path.join(G, "cornell.npz"))
    w, h = 160, 128
    with ra.Scene(sd) as sc:
        whole, _ = sc.render(w, h, 1, abi.RT_PHONG, abi.RT_JITTER_NONE)
        part, _ = sc.render(w, h, 1, abi.RT_PHONG, abi.RT_JITTER_NONE, tile_first=1, tile_stride=2)
        again, _ = sc.render(w, h, 1, abi.RT_PHONG, abi.RT_JITTER_NONE)
    tiles_x = (w + abi.RT_TILE - 1) // abi.RT_TILE
    for ty in range((h + abi.RT_TILE - 1) // abi.RT_TILE):
        for tx in range(tiles_x):
            block = part[ty * abi.RT_TILE:(ty + 1) * abi.RT_TILE, tx * abi.RT_TILE:(tx + 1) * abi.RT_TILE]
            ref = whole[ty * abi.RT_TILE:(ty + 1) * abi.RT_TILE, tx * abi.RT_TILE:(tx + 1) * abi.RT_TILE]
            if (ty * tiles_x + tx) % 2 == 1:
                assert np.array_equal(block, ref)
            else:
                assert not block.any()
    assert again.tobytes() == whole.tobytes()


def test_per_kernel_times_only_with_counters():
    sd = ra.SceneData.load(os.path.join(G, "balls2.npz"))
    with ra.Scene(sd) as sc:
        _, st = sc.render(200, 150, 1, abi.RT_PHONG, abi.RT_JITTER_HASH)
    assert st["ms_trace"] > 0 and st["ms_total"] > st["ms_trace"] and st["ms_light"] == 0 and st["ms_resolve"] == 0
    with ra.Scene(sd, counters=True) as sc:
        _, st = sc.render(200, 150, 1, abi.RT_PHONG, abi.RT_JITTER_HASH)
    assert st["ms_light"] > 0 and st["ms_combine"] > 0 and st["ms_resolve"] > 0
