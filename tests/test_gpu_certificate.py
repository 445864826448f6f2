"""The hit certificate changes no result, and is really used.

A closest-hit ray that finds, in the second tree, a nearest primitive with no rival within the tie window is answered
from the one root-to-leaf path of the reference's tree instead of the reference's whole walk.  These scenes go for the
conditions of that shortcut: layers of the same triangles stacked closer than, at, and just beyond EQUAL_EPSILON
(Util.h:6) along the view direction, exact duplicates, cameras inside the cloud of primitives (where the reference's
box test returns exit distances and may skip whole subtrees, kernel.cu:139-175), cameras a hair's breadth from a
surface, and crowded random scenes with mirrors and glass whose bounce rays start on surfaces.  Every frame, every
sample colour and every primary hit must be the same bits with the second tree and without it; against the CPU oracle
the primary hits must be the same bits too.
"""
import os

import numpy as np
import pytest

import orclib
import raytracer_amd as ra
from raytracer_amd import abi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def small_scenes_get_the_second_tree():
    old = os.environ.get("RT_FILTER_MIN_PRIMS")
    os.environ["RT_FILTER_MIN_PRIMS"] = "2"
    yield
    if old is None:
        os.environ.pop("RT_FILTER_MIN_PRIMS", None)
    else:
        os.environ["RT_FILTER_MIN_PRIMS"] = old


def both_ways(sd, w, h, spp=1, shading=abi.RT_PHONG, seed=7, oracle=True):
    got = []
    for use in (True, False):
        with ra.Scene(sd, miss_filter=use, counters=True) as sc:
            frame, st = sc.render(w, h, spp, shading, abi.RT_JITTER_HASH, seed)
            prim, t = sc.trace_primary(w, h, spp, abi.RT_JITTER_HASH, seed)
            rgb, _ = sc.render_samples(w, h, spp, shading, abi.RT_JITTER_HASH, seed)
            got.append((frame, prim, t, rgb, st))
    (f1, p1, t1, c1, s1), (f0, p0, t0, c0, s0) = got
    assert np.array_equal(p1, p0)
    assert np.array_equal(t1.view(np.uint32), t0.view(np.uint32))
    assert np.array_equal(c1.view(np.uint32), c0.view(np.uint32))
    assert f1.tobytes() == f0.tobytes()
    for k in ("rays_primary", "rays_shadow", "rays_reflect", "rays_refract", "shade_nodes", "shadow_skipped"):
        assert s1[k] == s0[k], k
    assert s0["hits_certified"] == 0 and s0["hits_contested"] == 0 and s0["hits_unreached"] == 0 and s0["shadows_certified"] == 0
    if oracle:
        osc = orclib.OracleScene(sd)
        try:
            r = osc.render(w, h, spp, shading, abi.RT_JITTER_HASH, seed, 0, 4, want_samples=False, want_primary=True)
        finally:
            osc.close()
        assert np.array_equal(r["prim"], p1)
        assert np.array_equal(r["t"].view(np.uint32), t1.view(np.uint32))
    return s1


def layered(offsets, n=60, seed=3, glass=False):
    """n random triangles facing the camera, and a copy of all of them at each offset along the view direction."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = ra.SceneData()
    base = ((rng.random((n, 3)) - 0.5) * np.array([6.0, 4.0, 1.0])).astype(np.float32)
    p2 = base + ((rng.random((n, 3)) - 0.5) * np.array([2.0, 2.0, 0.2])).astype(np.float32)
    p3 = base + ((rng.random((n, 3)) - 0.5) * np.array([2.0, 2.0, 0.2])).astype(np.float32)
    layers = [0.0] + list(offsets)
    tr = np.zeros(n * len(layers), abi.triangle_dtype)
    mod = synth.identity_modifier(len(tr))
    mod["mat"]["clr"] = rng.random((len(tr), 3)).astype(np.float32)
    mod["mat"]["amb"], mod["mat"]["dif"], mod["mat"]["spec"], mod["mat"]["rough"] = 0.2, 0.6, 0.3, 0.1
    if glass:
        mod["mat"]["refl"] = np.where(rng.random(len(tr)) < 0.5, 0.4, 0.0)
        mod["mat"]["refr"] = np.where(rng.random(len(tr)) < 0.3, 1.0, 0.0)
        mod["mat"]["alpha"] = np.where(mod["mat"]["refr"] > 0, 0.6, 0.0)
        mod["mat"]["ior"] = 1.3
    for k, dz in enumerate(layers):
        sl = slice(k * n, (k + 1) * n)
        shift = np.array([0.0, 0.0, dz], np.float32)
        tr["p1"][sl], tr["p2"][sl], tr["p3"][sl] = base + shift, p2 + shift, p3 + shift
    tr["mod"] = mod
    sd.triangles = tr
    lt = np.zeros(2, abi.light_dtype)
    lt["pos"], lt["clr"] = [(3.0, 5.0, 9.0), (-4.0, 2.0, 6.0)], (0.7, 0.7, 0.7)
    sd.lights = lt
    sd.camera = {"pos": np.array([0, 0, 8], np.float32), "up": np.array([0, 1, 0], np.float32),
                 "right": np.array([1.3333, 0, 0], np.float32), "look_at": np.array([0, 0, 0], np.float32)}
    return sd


@pytest.mark.parametrize("offsets", [(0.0,), (0.004,), (0.0095, -0.0095), (0.0101, 0.0205), (0.05, 0.3), (0.0, 0.0099, 0.0101, 1.0)])
def test_layers_around_the_tie_window(offsets):
    st = both_ways(layered(offsets), 240, 160, 1)
    assert st["hits_certified"] + st["hits_contested"] + st["hits_unreached"] > 0
    assert st["hits_guided"] <= st["hits_contested"]
    if min(abs(o) for o in offsets) < 0.009:
        assert st["hits_contested"] > 0          # rivals within the window: the reference's walk has to be made,
        assert st["hits_guided"] > 0             # but only where the few primitives near the hit are
    else:
        assert st["hits_certified"] > 0


# ---- contested hits: the guided walk -----------------------------------------------------------------------------------
# A ray with rivals within the tie window walks the reference's tree, but only the subtrees that hold one of the primitives
# it met in the second tree -- if those are few (four at most) and all within 0.03 of the nearest, while everything else is
# known to be 0.045 or more behind it.  The layers below sit on both sides of each of those limits.
@pytest.mark.parametrize("offsets", [(0.002, 0.004, 0.006), (0.002, 0.004, 0.006, 0.008), (0.001, 0.002, 0.003, 0.004, 0.005, 0.006),
                                     (0.004, 0.029), (0.004, 0.031), (0.004, 0.0295, 0.0305, 0.044, 0.046), (0.009, 0.018, 0.027, 0.036, 0.045, 0.054),
                                     (0.0, 0.0, 0.0), (0.0, 0.0, 0.0, 0.0, 0.0)])
def test_layers_around_the_limits_of_the_guided_walk(offsets):
    st = both_ways(layered(offsets, n=50, seed=11), 240, 160, 1)
    assert st["hits_contested"] > 0 and st["hits_guided"] <= st["hits_contested"]
    if len(offsets) <= 3 and max(offsets) < 0.028:
        assert st["hits_guided"] > 0
    if len(offsets) >= 5 and max(offsets) < 0.01:
        assert st["hits_guided"] < st["hits_contested"]        # more rivals than a ray keeps: the walk in full


def test_guided_walk_on_a_mesh_seen_at_its_vertices():
    """A height field looked at without jitter from straight above its middle: rays through vertices and along edges meet
    up to six triangles at one parameter."""
    sd = synth.terrain(24, 12, lights_per_side=1)
    sd.camera = {"pos": np.array([0.0, 30.0, 0.0], np.float32), "up": np.array([0, 0, 1], np.float32),
                 "right": np.array([1.3333, 0, 0], np.float32), "look_at": np.array([0.0, 0.0, 0.0], np.float32)}
    got = []
    for use in (True, False):
        with ra.Scene(sd, miss_filter=use, counters=True) as sc:
            frame, st = sc.render(193, 145, 1, abi.RT_PHONG, abi.RT_JITTER_NONE, 0)
            prim, t = sc.trace_primary(193, 145, 1, abi.RT_JITTER_NONE, 0)
            got.append((frame, prim, t, st))
    assert np.array_equal(got[0][1], got[1][1]) and np.array_equal(got[0][2].view(np.uint32), got[1][2].view(np.uint32))
    assert got[0][0].tobytes() == got[1][0].tobytes()
    osc = orclib.OracleScene(sd)
    try:
        r = osc.render(193, 145, 1, abi.RT_PHONG, abi.RT_JITTER_NONE, 0, 0, 4, want_samples=False, want_primary=True)
    finally:
        osc.close()
    assert np.array_equal(r["prim"], got[0][1]) and np.array_equal(r["t"].view(np.uint32), got[0][2].view(np.uint32))


def test_no_guided_walk_from_inside_the_tree():
    """A ray that starts inside the box of the reference's root may start inside boxes further down; their values are then
    exit distances and the argument for leaving subtrees out does not hold: such rays make the walk in full."""
    sd = layered((0.004, 0.006), n=50, seed=5)
    sd.camera["pos"] = np.array([0.0, 0.0, 0.3], np.float32)
    sd.camera["look_at"] = np.array([0.0, 0.0, -1.0], np.float32)
    st = both_ways(sd, 200, 150, 1)
    assert st["hits_guided"] == 0


def test_layers_with_mirrors_and_glass():
    both_ways(layered((0.0095, 0.02, 0.5), glass=True), 200, 140, 4)


@pytest.mark.parametrize("seed", range(6))
def test_crowded_random_scenes(seed):
    sd = synth.random_scene(100 + seed, n_spheres=120, n_triangles=150, n_boxes=40, n_smooth=60, n_planes=seed % 2, n_lights=2)
    st = both_ways(sd, 200, 150, 1)
    assert st["hits_certified"] > 0


@pytest.mark.parametrize("pos", [(0.0, 0.0, 0.0), (0.3, -0.2, 0.9), (1.5, 1.0, -1.0)])
def test_camera_inside_the_cloud(pos):
    """Rays that start inside boxes of the reference's tree: its box test gives exit distances there and its walk may
    skip subtrees; the certificate must notice that the leaf is not provably reached and make the walk."""
    sd = synth.random_scene(42, n_spheres=150, n_triangles=150, n_boxes=30, n_smooth=30, n_planes=0, n_lights=2)
    sd.camera = {"pos": np.array(pos, np.float32), "up": np.array([0, 1, 0], np.float32),
                 "right": np.array([1.3333, 0, 0], np.float32), "look_at": np.array([pos[0] + 0.3, pos[1] - 0.1, pos[2] - 1.0], np.float32)}
    st = both_ways(sd, 200, 150, 1)
    assert st["hits_unreached"] + st["hits_contested"] > 0


def test_camera_a_hair_from_a_surface():
    """Nearest hits below 2 * EQUAL_EPSILON are never certified (a primitive behind the origin could tie with them)."""
    sd = layered((0.5,), n=40)
    sd.camera["pos"] = np.array([0.0, 0.0, 0.515], np.float32)       # the second layer lies at z in [0, 1]: some of it within 0.02 of the image plane's rays
    sd.camera["look_at"] = np.array([0.0, 0.0, -1.0], np.float32)
    both_ways(sd, 200, 150, 1)


def test_full_frame_balls2_counts():
    """On the sphere flake most nearest hits are certified; the rest is split between rivals (touching spheres) and
    leaves that are not provably reached (bounce rays that start inside the flake's boxes)."""
    sd = ra.SceneData.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scenes", "balls2.npz"))
    st = both_ways(sd, 480, 270, 1, oracle=False)
    assert st["hits_certified"] > 5 * (st["hits_contested"] + st["hits_unreached"]) // 4


# ---- feelers: "shadowed" from the second tree -------------------------------------------------------------------------
# A feeler's parameter runs from 0 at the light to about 1 at its target, so EPSILON (1e-4) and EQUAL_EPSILON (1e-2) are
# fractions of the light's distance.  With the lights of layered() about 10 away, a copy of the layer 0.0005 in front of
# a point lies in the last EPSILON before it (the reference's walk stops there and says "lit"), one 0.05 in front within
# EQUAL_EPSILON (a tie with the point's own triangle may make the walk pass it by), one 0.3 or 1.0 in front is clear of
# both: only those may be answered without the reference's walk.  (The triangles of a layer also shadow each other, from
# further away.)
@pytest.mark.parametrize("offsets", [(0.0005,), (0.0005, 0.3), (0.05,), (0.05, 0.3), (0.3, 1.0), (0.0009, 0.0011, 0.09, 0.11)])
def test_feelers_and_layers_around_their_target(offsets):
    st = both_ways(layered(offsets), 240, 160, 1)
    assert st["shadows_certified"] > 0


def test_feelers_from_a_distant_light_are_never_certified():
    """From 1000 away everything near the target is within EQUAL_EPSILON of it."""
    sd = layered((0.3, 1.0))
    sd.lights["pos"] = [(300.0, 500.0, 900.0), (-400.0, 200.0, 600.0)]
    st = both_ways(sd, 240, 160, 1)
    assert st["shadows_certified"] == 0 and st["rays_shadow"] > 0


@pytest.mark.parametrize("seed,scale", [(77, 20.0), (78, 60.0), (79, 200.0), (80, 60.0), (81, 60.0), (81, 1000.0)])
def test_lights_far_from_small_spheres(seed, scale):
    """Lights 100 to 20 000 units from a scene of unit-sized spheres, boxes and triangles.  Sphere::intersects
    (Sphere.h:42-66) subtracts two products of the size (distance)^2, so from that far it reports its hits off the sphere by
    a good part of the radius, in front of the sphere's own box as often as not; the reference's walk takes them as they
    come, and a feeler's walk of the second tree has to look that much further than its target.  (With the margin of a
    nearby light, 1 to 30 of 480 000 samples came out lit where the reference has them shadowed.)"""
    sd = synth.random_scene(seed, n_spheres=150, n_triangles=150, n_boxes=40, n_smooth=60, n_planes=0, n_lights=2)
    sd.lights["pos"] = sd.lights["pos"] * scale
    both_ways(sd, 400, 300, 4, oracle=scale < 100)


def test_feelers_certified_in_crowded_scenes_and_on_the_flake():
    sd = synth.random_scene(321, n_spheres=150, n_triangles=150, n_boxes=40, n_smooth=60, n_planes=1, n_lights=3)
    st = both_ways(sd, 200, 150, 4)
    assert st["shadows_certified"] > 0
    sd = ra.SceneData.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scenes", "balls2.npz"))
    st = both_ways(sd, 320, 180, 1, oracle=False)
    assert st["shadows_certified"] > st["rays_shadow"] // 10


def test_wave_steps_are_split_by_tree():
    """RT_SCENE_COUNTERS: the wave node steps with rays in both trees and in the second tree only are counted, and a scene
    without the second tree has neither."""
    sd = ra.SceneData.load(os.path.join(os.path.dirname(__file__), "golden", "scenes", "balls2.npz"))
    with ra.Scene(sd, counters=True) as sc:
        for _ in range(7):
            _, st = sc.render(320, 240, 1)
    assert st["second_tree"] != 0
    assert 0 < st["wave_steps_both_trees"] and 0 < st["wave_steps_second_tree_only"]
    assert st["wave_steps_both_trees"] + st["wave_steps_second_tree_only"] <= st["wave_node_steps"]
    with ra.Scene(sd, counters=True, miss_filter=False) as sc:
        _, st = sc.render(320, 240, 1)
    assert st["second_tree"] == 0 and st["wave_steps_both_trees"] == 0 and st["wave_steps_second_tree_only"] == 0
