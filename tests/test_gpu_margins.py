"""The shortcuts of the traversal kernel against the reference's plain walk where their margins are tightest.

The second tree is only ever used to prove what the reference's walk (kernel.cu:83-200) returns; the proofs lean on
margins for the rounding of the box tests and of the primitive tests (the widening of the second tree's boxes, per box
and per ray; how far beyond a hit or a feeler's target it still looks).  These scenes go where those roundings are
large against the geometry: scenes far from the origin of the coordinates, scenes a thousand times smaller and larger,
lights and cameras far from small primitives, cameras a hair above large triangles, and rays that start inside boxes.
Every frame must be the same bytes, with the same ray counts, with every shortcut on (second tree, tiles left out,
feelers that cannot matter, tile order) and with all of them off (every ray makes the reference's walk); a subset is
also compared with the CPU oracle.
"""
import os

import numpy as np
import pytest

import orclib
import raytracer_amd as ra
from raytracer_amd import abi, synth

pytestmark = pytest.mark.gpu

OFF = {"RT_MISS_FILTER": "0", "RT_CULL_TILES": "0", "RT_SKIP_UNLIT": "0", "RT_TILE_ORDER": "0"}
COUNTS = ("rays_total", "rays_primary", "rays_shadow", "rays_reflect", "rays_refract", "shade_nodes", "shadow_skipped")
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "scenes")


@pytest.fixture(autouse=True)
def small_scenes_get_the_second_tree():
    old = os.environ.get("RT_FILTER_MIN_PRIMS")
    os.environ["RT_FILTER_MIN_PRIMS"] = "4"
    yield
    if old is None:
        os.environ.pop("RT_FILTER_MIN_PRIMS", None)
    else:
        os.environ["RT_FILTER_MIN_PRIMS"] = old


def render(sd, env, w, h, spp, shading, frames=2):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        with ra.Scene(sd, miss_filter=None if env else True) as sc:
            for _ in range(frames):
                f, st = sc.render(w, h, spp, shading, abi.RT_JITTER_HASH, 3)
            rgb, _ = sc.render_samples(w, h, spp, shading, abi.RT_JITTER_HASH, 3)
            st["filter_depth"] = sc.info()["filter_depth"]
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return f, rgb, st


def on_and_off(sd, w=160, h=120, spp=1, shading=abi.RT_PHONG, oracle=False):
    f1, c1, s1 = render(sd, {}, w, h, spp, shading)
    f0, c0, s0 = render(sd, OFF, w, h, spp, shading, frames=1)
    assert np.array_equal(c1.view(np.uint32), c0.view(np.uint32)), f"{int((c1.view(np.uint32) != c0.view(np.uint32)).any(axis=2).sum())} samples differ"
    assert f1.tobytes() == f0.tobytes()
    for k in COUNTS:
        assert s1[k] == s0[k], k
    assert s1["second_tree"] != 0 and s1["filter_depth"] > 0 and s0["filter_depth"] == 0
    if oracle:
        osc = orclib.OracleScene(sd)
        try:
            r = osc.render(w, h, spp, shading, abi.RT_JITTER_HASH, 3, 0, 4, want_samples=False, want_primary=False)
        finally:
            osc.close()
        diff = np.abs(f1[:, :, :3].astype(int) - r["rgba"][:, :, :3].astype(int)).max(axis=2)
        assert (diff <= 1).mean() >= 0.999
        c = r["counters"]
        assert (s1["rays_reflect"], s1["rays_refract"], s1["shade_nodes"]) == (c["rays_reflect"], c["rays_refract"], c["shade_nodes"])
    return s1


def crowd(seed, planes=0):
    return synth.random_scene(seed, n_spheres=60, n_triangles=80, n_boxes=20, n_smooth=40, n_planes=planes, n_lights=2)


@pytest.mark.parametrize("shift", [3.0e3, 2.0e4, 1.5e5])
@pytest.mark.parametrize("seed", [11, 12])
def test_scenes_far_from_the_origin(seed, shift):
    """Everything (objects, lights, camera) moved along a diagonal: coordinates carry 2e-4 to 1.6e-2 of rounding there."""
    sd = synth.similar(crowd(seed, planes=seed % 2), 1.0, (shift, -0.7 * shift, 0.4 * shift))
    on_and_off(sd, oracle=shift < 1e5)


@pytest.mark.parametrize("scale", [1.0e-3, 1.0e3])
@pytest.mark.parametrize("seed", [21, 22])
def test_scenes_a_thousand_times_smaller_and_larger(seed, scale):
    """The whole scene scaled, camera and lights with it: EPSILON and EQUAL_EPSILON (Util.h:4-6) stay what they are, so at
    1e-3 every hit ties with its neighbours and at 1e3 nothing does."""
    on_and_off(synth.similar(crowd(seed), scale), oracle=True)


@pytest.mark.parametrize("scale,seed", [(1.0e-3, 31), (1.0e-2, 32), (1.0e-1, 33)])
def test_small_primitives_seen_and_lit_from_afar(scale, seed):
    """The objects shrunk, lights left where they are and the camera too, looking at them through a narrow view: ray origins
    10^2 to 10^4 primitive sizes away, where the sphere test's hits wander off their spheres."""
    sd = synth.similar(crowd(seed), scale, move_camera=False, move_lights=False)
    cam = sd.camera
    sd.camera = dict(cam, up=np.asarray(cam["up"], np.float32) * np.float32(scale), right=np.asarray(cam["right"], np.float32) * np.float32(scale),
                     look_at=np.asarray(cam["look_at"], np.float32) * np.float32(scale))
    on_and_off(sd, 200, 150)


@pytest.mark.parametrize("height", [1.0e-3, 1.0e-2, 0.2])
def test_camera_a_hair_above_large_triangles(height):
    """Two ground triangles 40 across and a crowd standing on them; the camera skims the ground, so most primaries and many
    feelers meet the ground's plane at a grazing angle (t = numer / denom with a tiny denominator, Triangle.h:60-70)."""
    sd = crowd(41)
    g = np.zeros(2, abi.triangle_dtype)
    g["mod"] = synth.identity_modifier(2)
    g["mod"]["mat"]["clr"], g["mod"]["mat"]["refl"] = (0.6, 0.6, 0.5), 0.3
    y = -2.0
    g["p1"], g["p2"], g["p3"] = [(-20, y, -20), (-20, y, -20)], [(20, y, -20), (20, y, 20)], [(20, y, 20), (-20, y, 20)]
    sd.triangles = np.concatenate([sd.triangles, g])
    sd.camera = {"pos": np.array([0.3, y + height, 9.0], np.float32), "up": np.array([0, 1, 0], np.float32),
                 "right": np.array([1.3333, 0, 0], np.float32), "look_at": np.array([0.0, y + height, 0.0], np.float32)}
    on_and_off(sd, 200, 150, oracle=True)


@pytest.mark.parametrize("seed", [51, 52, 53])
def test_rays_that_start_inside_boxes(seed):
    """Camera in the middle of a crowd of glass and mirrors: primaries and bounce rays start inside boxes of both trees,
    where the reference's box test returns exit distances (GeometryUtil.h:41-47)."""
    sd = synth.glassy(crowd(seed), everything=True)
    rng = np.random.Generator(np.random.PCG64(seed))
    pos = (rng.random(3) - 0.5).astype(np.float32)
    sd.camera = {"pos": pos, "up": np.array([0, 1, 0], np.float32), "right": np.array([1.3333, 0, 0], np.float32),
                 "look_at": (pos + np.array([0.2, -0.1, -1.0], np.float32)).astype(np.float32)}
    on_and_off(sd, 120, 90, oracle=True)


def test_balls2_lit_from_afar():
    """The sphere flake (spheres down to radius 0.006) with its lights ten and a hundred times further away."""
    sd = ra.SceneData.load(os.path.join(GOLDEN, "balls2.npz"))
    for far in (10.0, 100.0):
        moved = synth.similar(sd, 1.0, move_camera=False, move_lights=False)
        moved.lights["pos"] = sd.lights["pos"] * np.float32(far)
        on_and_off(moved, 320, 180)


@pytest.mark.parametrize("far", [50.0, 3.0e3, 2.0e5])
def test_camera_moved_beyond_the_reach_the_second_tree_was_built_for(far):
    """The second tree's boxes are widened for ray origins up to twice as far from the origin of the coordinates as the
    scene, its lights and its first camera reach; a camera moved further out afterwards sends rays that must leave the
    tree alone.  The frames from out there are the same bytes with the second tree and without it."""
    sd = synth.random_scene(91, n_spheres=60, n_triangles=80, n_boxes=20, n_smooth=30, n_planes=0, n_lights=2)
    frames = []
    for use in (True, False):
        with ra.Scene(sd, miss_filter=use, counters=True) as sc:
            near, st_near = sc.render(160, 120, 1, abi.RT_PHONG, abi.RT_JITTER_HASH, 3)
            pos = np.array([0.3 * far, 0.2 * far, far], np.float32)
            # (seen from that far the scene is a speck: a narrow image plane keeps it in the picture)
            sc.set_camera(pos, np.array([0, 8.0 / far, 0], np.float32), np.array([10.6 / far, 0, 0], np.float32), np.zeros(3, np.float32))
            out, st_out = sc.render(160, 120, 1, abi.RT_PHONG, abi.RT_JITTER_HASH, 3)
            prim, t = sc.trace_primary(160, 120, 1, abi.RT_JITTER_HASH, 3)
            frames.append((near, out, prim, t, st_near, st_out))
    (n1, o1, p1, t1, sn1, so1), (n0, o0, p0, t0, sn0, so0) = frames
    assert n1.tobytes() == n0.tobytes() and o1.tobytes() == o0.tobytes()
    assert np.array_equal(p1, p0) and np.array_equal(t1.view(np.uint32), t0.view(np.uint32))
    assert (p1 >= 0).sum() > 200                      # the scene is in the picture
    for k in COUNTS:
        assert so1[k] == so0[k], k
    assert sn1["hits_certified"] > 0                  # near by, the second tree answers
    if far > 1.0e3:
        # from beyond the reach the primaries make the reference's walk (bounce rays start on the scene and still use the tree)
        assert so1["hits_certified"] < int((p1 >= 0).sum()) // 4
