"""How the oracle's composition -- traversal order, tie rule, shadeObject recursion, bounce and shadow decisions -- is
pinned beyond its leaves (which are bit-exact against the reference's own classes, test_oracle_goldens.py).

1. Ray counts.  A frame's ray count depends on every hit, miss, spawn and shadow decision of every pixel.  A host
   build of the reference's own kernel.cu logic, made before this code existed, recorded the counts of seven scenes
   (no jitter, Phong, 800x600; simple.pov 640x480).  The oracle reproduces all seven exactly when Sphere.h:55,63 is
   read as that host build read it (unqualified sqrt(float) bound to ::sqrt(double), the expression evaluated in
   double), and the five scenes without that ambiguity exactly either way.  With the binary32 reading that nvcc's
   device headers give (the one the oracle and the HIP path use) the two sphere scenes differ by -38 and +76 rays in
   2.3 and 7.4 million: rays whose t changes in the last bit.

2. The reference's own images, on interior pixels.  The .tga files were rendered by the author's CUDA build with
   cuRAND jitter, so silhouette and shadow-edge pixels cannot match.  A pixel is interior when, for each of its
   samples, the oracle's whole shade tree makes the same decisions for the unjittered sample and for eight draws on
   the boundary of the jitter square.  There the images must agree within 2/255.  bunny.tga and
   dragon_large_tris.tga agree best with a 9-sample render (98.98 % and 98.33 % of all pixels within 1/255, against
   96.74 % and 95.06 % at one sample): they were made with `-s 9` and are compared so.  What remains (1 interior pixel
   in 1 000 on the meshes, 4 in 10 000 on final1, none on recurse_simp2) are isolated lit/shadowed and front/back
   flips where two surfaces coincide, decisions that the reference's -use_fast_math build rounds differently.
"""
import os

import numpy as np
import pytest

import orclib
import raytracer_amd as ra
from raytracer_amd import abi

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# scene: (width, height, rays counted by the host-compiled reference logic, difference of the binary32 reading)
REFERENCE_RAY_COUNTS = {
    "simple": (640, 480, 467_423, 0),
    "valentine": (800, 600, 990_929, 0),
    "bunny_small_tris": (800, 600, 853_600, 0),
    "bunny": (800, 600, 539_070, 0),
    "dragon_large_tris": (800, 600, 853_600, 0),
    "balls2": (800, 600, 2_311_040, -38),
    "iceBot": (800, 600, 7_376_080, +76),
}


def scene(name):
    return ra.SceneData.load(os.path.join(GOLD, "scenes", name + ".npz"))


def ray_count(name, w, h):
    osc = orclib.OracleScene(scene(name))
    try:
        c = osc.render(w, h, 1, abi.RT_PHONG, abi.RT_JITTER_NONE, 0, 0, 0, want_samples=False, want_primary=False)["counters"]
    finally:
        osc.close()
    # every getClosestIntersection call, as the reference issues them: feelers whose own-surface test missed included
    return c["rays_primary"] + c["rays_shadow"] + c["rays_reflect"] + c["rays_refract"]


@pytest.mark.parametrize("name", sorted(REFERENCE_RAY_COUNTS))
def test_ray_counts_of_the_host_compiled_reference(name):
    w, h, want, float_residue = REFERENCE_RAY_COUNTS[name]
    assert ray_count(name, w, h) == want + float_residue
    if float_residue:
        orclib.orc().orc_set_sphere_sqrt_double(1)
        try:
            assert ray_count(name, w, h) == want
        finally:
            orclib.orc().orc_set_sphere_sqrt_double(0)


@pytest.mark.parametrize("name", ["cornell", "balls2", "bunny", "iceBot", "dragon_large_tris", "box", "final2", "transforms"])
def test_reference_tree_walk_against_brute_force(name):
    """The oracle's walk of the reference tree never loses a hit: against the nearest intersection found by testing every
    primitive and plane, hit / miss is the same on every primary ray and t lies within the tie window
    (EQUAL_EPSILON, Util.h:6; kernel.cu:108-122 may keep the farther of two hits that close) above the minimum."""
    osc = orclib.OracleScene(scene(name))
    try:
        prim_b, t_b = osc.brute_primary(160, 120)
        r = osc.render(160, 120, 1, abi.RT_PHONG, abi.RT_JITTER_NONE, 0, 0, 0, want_samples=False, want_primary=True)
    finally:
        osc.close()
    hit, hit_b = r["prim"] >= 0, prim_b >= 0
    assert np.array_equal(hit, hit_b)
    d = r["t"][hit] - t_b[hit]
    assert d.min() >= -1e-4 and d.max() < 0.01
    assert (r["prim"] == prim_b).mean() >= 0.99


# scene: (samples per pixel the image was made with, floor of interior pixels within 2/255, floor of the interior share)
INTERIOR = {
    "recurse_simp2": (1, 0.9999, 0.97),
    "valentine": (1, 0.9999, 0.99),
    "final1": (1, 0.9995, 0.99),
    "bunny": (9, 0.9985, 0.92),
    "dragon_large_tris": (9, 0.9982, 0.90),
}


@pytest.mark.parametrize("name", sorted(INTERIOR))
def test_interior_pixels_against_reference_tga(name):
    spp, floor, share = INTERIOR[name]
    want = np.load(os.path.join(GOLD, "tga", name + ".npz"))["rgb"]       # rows bottom -> top, RGB
    h, w = want.shape[:2]
    osc = orclib.OracleScene(scene(name))
    try:
        got = osc.render(w, h, spp, abi.RT_PHONG, abi.RT_JITTER_NONE, 0, 0, 0, want_samples=False, want_primary=False)["rgba"]
        interior = osc.interior_mask(w, h, spp)
    finally:
        osc.close()
    diff = np.abs(got[:, :, :3].astype(np.int16) - want.astype(np.int16)).max(axis=2)
    frac = float((diff[interior] <= 2).mean())
    print(f"{name}: {interior.mean():.4f} of the pixels are interior, {frac:.6f} of those within 2/255 ({int((diff[interior] > 2).sum())} are not)")
    assert interior.mean() >= share
    assert frac >= floor
