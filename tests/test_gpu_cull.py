"""Leaving whole tiles of primaries out of the launch changes no byte.

A 32x32-pixel tile through which no ray can meet anything (every ray passes the box that holds all primitives by a
margin; worked out on the host per camera and frame size, only for scenes without planes) is not traced at all.
Frames must be the same bytes with and without that (RT_CULL_TILES=0 turns it off, read when a scene is created):
on the fixture scenes, on random scenes seen from near, far and inside, with anti-aliasing, after the camera moves,
in tile parts; and the ray counts must not change, because the primaries that are left out are still answered.
"""
import os

import numpy as np
import pytest

import raytracer_amd as ra
from raytracer_amd import abi, synth

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scenes")


def scene_with(cull, sd, **kw):
    old = os.environ.get("RT_CULL_TILES")
    os.environ["RT_CULL_TILES"] = "1" if cull else "0"
    try:
        return ra.Scene(sd, **kw)
    finally:
        if old is None:
            os.environ.pop("RT_CULL_TILES", None)
        else:
            os.environ["RT_CULL_TILES"] = old


def both(sd, w, h, spp, shading, n=1, **kw):
    out = []
    for cull in (True, False):
        with scene_with(cull, sd, **kw) as sc:
            last = None
            for _ in range(n):
                last, st = sc.render(w, h, spp, shading, abi.RT_JITTER_HASH, 5)
            out.append((last, st))
    return out


def look_from(sd, pos, look_at=(0.0, 0.0, 0.0), right=1.3333):
    sd.camera = {"pos": np.array(pos, np.float32), "up": np.array([0, 1, 0], np.float32),
                 "right": np.array([right, 0, 0], np.float32), "look_at": np.array(look_at, np.float32)}
    return sd


@pytest.mark.parametrize("name", ["bunny", "dragon_large_tris", "balls2", "iceBot", "valentine", "recurse_simp2", "cornell", "blitzcrank",
                                  "gnarly", "miniShark", "monkey", "blitz", "test2", "test3"])      # (the last six: scenes without planes, where tiles can be left out)
def test_fixture_scenes_same_bytes(name):
    path = os.path.join(G, name + ".npz")
    if not os.path.exists(path):
        pytest.skip("no such fixture scene")
    sd = ra.SceneData.load(path)
    (f1, s1), (f0, s0) = both(sd, 333, 257, 1, abi.RT_PHONG, n=2)     # edges cut the last tile column and row; the second frame has a start order
    assert f1.tobytes() == f0.tobytes()
    assert s0["primaries_culled"] == 0
    for k in ("rays_total", "rays_primary", "rays_shadow", "rays_reflect", "rays_refract", "shade_nodes"):
        assert s1[k] == s0[k], k
    if len(sd.planes) > 0:
        assert s1["primaries_culled"] == 0       # planes are tested for every ray


def test_bunny_headline_shape_culls_most_of_the_frame():
    sd = ra.SceneData.load(os.path.join(G, "bunny.npz"))
    (f1, s1), (f0, s0) = both(sd, 640, 360, 4, abi.RT_COOK_TORRANCE, n=3)
    assert f1.tobytes() == f0.tobytes()
    assert s1["primaries_culled"] > s1["rays_primary"] // 3
    assert s1["rays_total"] == s0["rays_total"]


@pytest.mark.parametrize("seed", range(8))
def test_random_scenes_near_far_and_inside(seed):
    sd = synth.random_scene(100 + seed, n_planes=0)
    rng = np.random.Generator(np.random.PCG64(seed))
    views = [((0, 1, 7), (0, 0, 0)), ((0, 2, 25), (0, 0, 0)), ((12, 3, 9), (1.5, 0, 0)), ((0.1, 0.1, 0.2), (0, 0, -1)),
             ((0, 1, 7), (6, 0, 0)),                                     # the scene in a corner of the frame
             (tuple(rng.normal(size=3) * 9), tuple(rng.normal(size=3)))]
    for pos, at in views:
        look_from(sd, pos, at)
        for (w, h, spp) in ((200, 150, 4), (417, 233, 1)):
            (f1, s1), (f0, s0) = both(sd, w, h, spp, abi.RT_COOK_TORRANCE if seed & 1 else abi.RT_PHONG)
            assert f1.tobytes() == f0.tobytes(), (pos, at, w, h, spp)
            assert s1["rays_total"] == s0["rays_total"]
    look_from(sd, (0.1, 0.1, 0.2), (0, 0, -1))
    with scene_with(True, sd) as sc:
        _, st = sc.render(200, 150, 1, abi.RT_PHONG, abi.RT_JITTER_HASH, 5)
    assert st["primaries_culled"] == 0           # the camera is inside the scene's box: nothing can be left out


def test_looking_away_leaves_every_tile_out():
    """No tile is live: no traversal launch at all, the frame is background, the primaries are still counted."""
    sd = look_from(synth.random_scene(3, n_planes=0), (0, 1, 40), (0, 1, 100))
    (f1, s1), (f0, s0) = both(sd, 200, 150, 4, abi.RT_PHONG, n=2)
    assert f1.tobytes() == f0.tobytes()
    assert s1["primaries_culled"] == s1["rays_primary"] == 200 * 150 * 4
    assert s1["rays_total"] == s0["rays_total"] == s1["rays_primary"]
    assert not f1[..., :3].any()


@pytest.mark.parametrize("shift", [3.0e3, 2.0e4, 1.5e5])
def test_far_from_the_origin_of_the_coordinates(shift):
    """Scene and camera moved far out: the float arithmetic of the ray set-up gets coarse (at 1.5e5 a unit in the last
    place is 1/64), and the margins of the tile test have to grow with it."""
    sd = synth.random_scene(11, n_spheres=0, n_boxes=0, n_planes=0, n_triangles=40, n_smooth=20)
    off = np.array([shift, -0.5 * shift, 0.25 * shift], np.float32)
    for arr in (sd.triangles, sd.smooth_triangles):
        arr["mod"]["trans"][:] = np.eye(4, dtype=np.float32).reshape(16)
        arr["mod"]["inv_trans"][:] = np.eye(4, dtype=np.float32).reshape(16)
        for k in ("p1", "p2", "p3"):
            arr[k] += off
    sd.lights["pos"] += off
    for pos, at in (((0, 1, 9), (0, 0, 0)), ((14, 2, 9), (5, 0, 0)), ((0, 1, 9), (0, 1, 30))):
        look_from(sd, np.array(pos, np.float32) + off, np.array(at, np.float32) + off)
        (f1, s1), (f0, s0) = both(sd, 320, 200, 4, abi.RT_PHONG)
        assert f1.tobytes() == f0.tobytes(), (shift, pos)
        assert s1["rays_total"] == s0["rays_total"]


def test_small_objects_on_tile_borders():
    """Spheres a pixel or two across, centred on tile corners of the frame: a tile is left out only if nothing reaches into it."""
    w, h = 512, 384
    sd = synth.random_scene(7, n_spheres=4, n_triangles=0, n_boxes=0, n_smooth=0, n_planes=0, n_lights=1)
    look_from(sd, (0, 0, 10), (0, 0, 0), right=4.0 / 3.0)
    # image plane through pos spans u/2*right, v/2*up; a point at depth 10 from E' = pos - look_dir (11 from E) projects by similar triangles
    sp = np.zeros(4, abi.sphere_dtype)
    sp["mod"] = sd.spheres["mod"][:4]
    sp["mod"]["trans"][:] = np.eye(4, dtype=np.float32).reshape(16)
    sp["mod"]["inv_trans"][:] = np.eye(4, dtype=np.float32).reshape(16)
    corners = [(160, 96), (352, 288), (192, 256), (320, 128)]      # the box around them ends on tile borders
    for k, (px, py) in enumerate(corners):
        u, v = 2.0 * px / w - 1.0, 2.0 * py / h - 1.0
        # direction of the line through E for (u, v): u/2 right + v/2 up + look_dir; the sphere sits 11 units along look_dir from E
        sp["c"][k] = (0.5 * u * (4.0 / 3.0) * 11.0, 0.5 * v * 11.0, 10.0 - 10.0)
        sp["r"][k] = 0.03 + 0.01 * k
    sd.spheres = sp
    (f1, s1), (f0, s0) = both(sd, w, h, 4, abi.RT_PHONG)
    assert f1.tobytes() == f0.tobytes()
    assert s1["primaries_culled"] > 0
    assert (f1[..., :3] != 0).any()              # the spheres are in the picture


def test_camera_moves_between_frames():
    sd = ra.SceneData.load(os.path.join(G, "bunny.npz"))
    cam0 = {k: np.array(v, np.float32) for k, v in sd.camera.items()}
    moved = dict(cam0)
    moved["pos"] = cam0["pos"] + np.array([3.0, 1.0, 0.0], np.float32)
    with scene_with(True, sd) as sc:
        for _ in range(3):
            a0, _ = sc.render(320, 200, 4, abi.RT_PHONG, abi.RT_JITTER_HASH, 5)
        sc.set_camera(moved["pos"], moved["up"], moved["right"], moved["look_at"])
        a1, st1 = sc.render(320, 200, 4, abi.RT_PHONG, abi.RT_JITTER_HASH, 5)
        a2, _ = sc.render(320, 200, 4, abi.RT_PHONG, abi.RT_JITTER_HASH, 5)
        sc.set_camera(cam0["pos"], cam0["up"], cam0["right"], cam0["look_at"])
        a3, _ = sc.render(320, 200, 4, abi.RT_PHONG, abi.RT_JITTER_HASH, 5)
    with scene_with(False, sd) as sc:
        b0, _ = sc.render(320, 200, 4, abi.RT_PHONG, abi.RT_JITTER_HASH, 5)
        sc.set_camera(moved["pos"], moved["up"], moved["right"], moved["look_at"])
        b1, _ = sc.render(320, 200, 4, abi.RT_PHONG, abi.RT_JITTER_HASH, 5)
    assert a0.tobytes() == b0.tobytes() and a3.tobytes() == b0.tobytes()
    assert a1.tobytes() == b1.tobytes() and a2.tobytes() == b1.tobytes()
    assert st1["primaries_culled"] > 0


def test_tile_parts():
    sd = ra.SceneData.load(os.path.join(G, "dragon_large_tris.npz"))
    w, h = 400, 300
    for first in range(3):
        outs = []
        for cull in (True, False):
            with scene_with(cull, sd) as sc:
                f, st = sc.render(w, h, 1, abi.RT_PHONG, abi.RT_JITTER_HASH, 5, 0, first, 3)
                f, st = sc.render(w, h, 1, abi.RT_PHONG, abi.RT_JITTER_HASH, 5, 0, first, 3)
                outs.append((f, st))
        assert outs[0][0].tobytes() == outs[1][0].tobytes()
        assert outs[0][1]["rays_total"] == outs[1][1]["rays_total"]


def test_finer_test_when_the_camera_stands_and_never_for_one_that_moves():
    """The quick test (the scene's box) is made for a new camera at once; the finer one, down the tree to the primitives'
    boxes, when the next frame comes with the same camera.  Sparse triangles in a wide box: the second frame leaves out
    more tiles than the first, every frame is the same bytes as without any culling, and a camera that moves every
    frame only ever gets the quick test."""
    sd = synth.random_scene(52, n_spheres=0, n_boxes=0, n_planes=0, n_triangles=25, n_smooth=10)
    look_from(sd, (0.0, 2.0, 30.0))
    w, h = 640, 480
    with scene_with(False, sd) as sc:
        want, st0 = sc.render(w, h, 1, abi.RT_PHONG, abi.RT_JITTER_HASH, 5)
    with scene_with(True, sd) as sc:
        culled = []
        for _ in range(4):
            got, st = sc.render(w, h, 1, abi.RT_PHONG, abi.RT_JITTER_HASH, 5)
            assert got.tobytes() == want.tobytes()
            for k in ("rays_total", "rays_primary", "rays_shadow", "shade_nodes"):
                assert st[k] == st0[k], k
            culled.append(st["primaries_culled"])
        assert culled[0] > 0 and culled[1] > culled[0] and culled[2] == culled[1] == culled[3]
        # a camera that never stands still
        quick = []
        for k in range(4):
            pos = np.array([0.01 * (k + 1), 2.0, 30.0], np.float32)
            sc.set_camera(pos, sd.camera["up"], sd.camera["right"], sd.camera["look_at"])
            _, st = sc.render(w, h, 1, abi.RT_PHONG, abi.RT_JITTER_HASH, 5)
            quick.append(st["primaries_culled"])
        assert max(quick) < culled[1]
    old = os.environ.get("RT_CULL_FINE")
    os.environ["RT_CULL_FINE"] = "0"
    try:
        with scene_with(True, sd) as sc:
            for _ in range(3):
                got, st = sc.render(w, h, 1, abi.RT_PHONG, abi.RT_JITTER_HASH, 5)
            assert got.tobytes() == want.tobytes() and st["primaries_culled"] == culled[0]
    finally:
        if old is None:
            os.environ.pop("RT_CULL_FINE", None)
        else:
            os.environ["RT_CULL_FINE"] = old
