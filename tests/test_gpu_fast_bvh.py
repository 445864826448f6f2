"""The opt-in surface-area-heuristic tree (RT_SCENE_FAST_BVH): same primitives, same kernels, another tree.

It is NOT a parity mode.  Two things in the reference depend on the order in which a tree visits primitives:
which of two hits within 0.01 of each other (in ray parameter) is reported (kernel.cu:108-122), and whether a
shadow feeler that meets an occluder within 0.01 of its own surface is reported shadowed (the early-out of
kernel.cu:117-121 against the self test of kernel.cu:351; the feeler's parameter is 1 at the surface, so the
window is 1 % of the distance to the light).  Where objects touch -- a sphere on a floor, the sphere flake of
balls2 -- contact shadows and the bounce rays that leave such points differ between any two trees.  What must
hold: primary hit/miss agrees, primary hit distances agree to within the tie window, primary ids agree on all
but a fraction of a percent, the pictures agree on the large majority of pixels, and the tree does far fewer box
tests.
"""
import os

import numpy as np
import pytest

import orclib
import raytracer_amd as ra
from raytracer_amd import abi, synth

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("builder", ["fast_bvh", "gpu_bvh"])
@pytest.mark.parametrize("name", ["cornell", "balls2", "bunny", "iceBot", "dragon_large_tris"])
def test_fast_tree_agrees_with_reference_tree(name, builder):
    sd = ra.SceneData.load(os.path.join(GOLDEN, "scenes", name + ".npz"))
    w, h = 320, 240
    with ra.Scene(sd, counters=True) as sc:
        prim0, t0 = sc.trace_primary(w, h)
        frame0, st0 = sc.render(w, h, 1, abi.RT_PHONG, abi.RT_JITTER_NONE)
    with ra.Scene(sd, counters=True, **{builder: True}) as sc:
        info = sc.info()
        prim1, t1 = sc.trace_primary(w, h)
        frame1, st1 = sc.render(w, h, 1, abi.RT_PHONG, abi.RT_JITTER_NONE)
    assert info["n_nodes"] == info["n_bvh_prims"] - 1
    hit0, hit1 = prim0 >= 0, prim1 >= 0
    assert (hit0 == hit1).mean() >= 0.9995
    both = hit0 & hit1
    assert np.abs(t0[both] - t1[both]).max() <= 0.0101          # the tie window (EQUAL_EPSILON, Util.h:6)
    assert (prim0 == prim1).mean() >= 0.995
    diff = np.abs(frame0[:, :, :3].astype(int) - frame1[:, :, :3].astype(int)).max(axis=2)
    agree = float((diff <= 1).mean())
    print(f"{name} [{builder}]: ids equal {(prim0 == prim1).mean():.4f}, pixels within 1/255 {agree:.4f}, "
          f"box-test records per ray {st0['node_visits'] / max(1, st0['rays_primary'] + st0['rays_reflect'] + st0['rays_refract']):.1f} -> "
          f"{st1['node_visits'] / max(1, st1['rays_primary'] + st1['rays_reflect'] + st1['rays_refract']):.1f}")
    assert agree >= 0.80
    visits0 = st0["node_visits"] + st0["node_visits_shadow"]
    visits1 = st1["node_visits"] + st1["node_visits_shadow"]
    assert visits1 < visits0


def check_against_brute_force(sd, w, h, builder):
    """Primary rays of a fast tree against the oracle's brute-force nearest hit (every primitive tested, no tree):
    hit / miss must agree on every sample and t must lie within the reference's tie window above the true minimum.
    A tree that lost a primitive, or fitted a box too small, fails this somewhere; a comparison with the product's own
    reference-tree output would not notice a fault both share."""
    osc = orclib.OracleScene(sd)
    try:
        prim_b, t_b = osc.brute_primary(w, h)
    finally:
        osc.close()
    with ra.Scene(sd, **{builder: True}) as sc:
        prim, t = sc.trace_primary(w, h)
    hit, hit_b = prim >= 0, prim_b >= 0
    assert np.array_equal(hit, hit_b), f"{int((hit != hit_b).sum())} samples disagree on hit / miss"
    d = t[hit] - t_b[hit]
    assert d.min() >= -1e-4 and d.max() <= 0.0101, (float(d.min()), float(d.max()))       # EQUAL_EPSILON, Util.h:6
    return float((prim == prim_b).mean())


@pytest.mark.parametrize("builder", ["fast_bvh", "gpu_bvh"])
@pytest.mark.parametrize("name", ["cornell", "balls2", "bunny", "iceBot", "dragon_large_tris", "box", "transforms"])
def test_fast_tree_against_brute_force(name, builder):
    """One scene at least per primitive class: spheres with transforms (balls2, transforms), boxes (box, cornell), flat
    triangles (bunny, dragon), smooth triangles (iceBot), planes (cornell, iceBot, dragon)."""
    sd = ra.SceneData.load(os.path.join(GOLDEN, "scenes", name + ".npz"))
    same = check_against_brute_force(sd, 160, 120, builder)
    print(f"{name} [{builder}]: nearest primitive equal on {same:.4f} of the samples")
    assert same >= 0.99


@pytest.mark.parametrize("builder", ["fast_bvh", "gpu_bvh"])
def test_fast_tree_of_a_million_triangles_against_brute_force(builder):
    sd = synth.terrain(1000, 500, lights_per_side=1)
    check_against_brute_force(sd, 48, 27, builder)


def test_fast_tree_makes_the_large_height_field_tractable():
    """200 000 triangles under 16 lights: the reference tree needs thousands of box tests per ray here."""
    sd = synth.terrain(400, 250, lights_per_side=4)
    with ra.Scene(sd, counters=True, fast_bvh=True) as sc:
        info = sc.info()
        frame, st = sc.render(640, 360, 1, abi.RT_PHONG, abi.RT_JITTER_HASH)
    n_rays = st["rays_total"]
    assert info["tree_depth"] <= 18 + 8
    assert (st["node_visits"] + st["node_visits_shadow"]) / n_rays < 200
    assert len(np.unique(frame[:, :, :3].reshape(-1, 3), axis=0)) > 1000


def test_device_build_of_a_large_scene():
    """A million triangles: the device builder takes milliseconds, and the picture matches the host SAH tree's."""
    sd = synth.terrain(1000, 500, lights_per_side=2)
    with ra.Scene(sd, gpu_bvh=True) as sc:
        info = sc.info()
        prim_g, t_g = sc.trace_primary(320, 180)
    assert info["n_nodes"] == 999999 and 0 < info["ms_build_device"] < 100.0
    with ra.Scene(sd, fast_bvh=True) as sc:
        prim_s, t_s = sc.trace_primary(320, 180)
    assert ((prim_g >= 0) == (prim_s >= 0)).mean() >= 0.9995
    both = (prim_g >= 0) & (prim_s >= 0)
    assert np.abs(t_g[both] - t_s[both]).max() <= 0.0101
