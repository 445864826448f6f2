"""Pins the CPU oracle (oracle/rt_oracle.c) against the reference's OWN classes.

oracle/_ref/libref.so is the reference's Sphere/Plane/Box/Triangle/
SmoothTriangle/BoundingBox/GeometryUtil/Phong/Cook-Torrance/PointLight code
compiled with g++ from /root/reference (recipe: oracle/Makefile).  Every
comparison here is BIT-EXACT (same IEEE binary32 operations in the same order).
Skipped where the library has not been built (it cannot be built without the
reference sources).
"""
import ctypes as C
import glob
import os

import numpy as np
import pytest

import orclib
from raytracer_amd import abi

pytestmark = pytest.mark.skipif(not orclib.have_ref(), reason="oracle/_ref/libref.so not built")

GOLDEN_SCENES = os.path.join(os.path.dirname(__file__), "golden", "scenes")
fp = orclib.fp


def bits(a):
    return np.asarray(a, dtype=np.float32).view(np.uint32)


def scene_paths():
    names = ["simple", "transforms", "box", "cornell", "final2", "valentine", "simple_tri", "recurse_simp2",
             "simple_refract", "ugly_part", "bunny_small_tris", "iceBot", "balls2"]
    out = []
    for n in names:
        p = os.path.join(orclib.REFERENCE_DIR, n + ".pov")
        if os.path.exists(p):
            out.append(p)
    return out


def rays_for_scene(rng, sd, osc, n):
    """Rays aimed at primitive centres (so most of them hit something) plus pure noise."""
    rays = np.empty((n, 6), np.float32)
    nprim = osc.n_bvh_prims + osc.n_planes
    c = np.zeros(3, np.float32)
    for i in range(n):
        o = rng.normal(0, 6, 3).astype(np.float32)
        if i % 4 != 3 and osc.n_bvh_prims:
            orclib.orc().orc_prim_center(osc.h, int(rng.integers(0, osc.n_bvh_prims)), c.ctypes.data_as(fp))
            d = (c + rng.normal(0, 0.05, 3).astype(np.float32)) - o
        else:
            d = rng.normal(0, 1, 3).astype(np.float32)
        if i % 7 == 0:
            d[int(rng.integers(0, 3))] = 0.0       # axis-parallel rays exercise the isFloatZero branches
        rays[i, :3], rays[i, 3:] = o, d
    return rays, nprim


@pytest.mark.parametrize("path", scene_paths(), ids=lambda p: os.path.basename(p))
def test_primitives_bit_exact(path):
    sd = orclib.ref_parse(path)
    osc = orclib.OracleScene(sd)
    rs = orclib.ref().ref_scene_create(C.byref(osc.desc))
    try:
        rng = np.random.default_rng(1234)
        nprim = osc.n_bvh_prims + osc.n_planes
        prims = np.arange(nprim) if nprim <= 400 else rng.choice(nprim, 400, replace=False)
        a3, b3, c3, d3 = (np.zeros(3, np.float32) for _ in range(4))
        # bounding boxes and sort keys
        for p in prims:
            p = int(p)
            orclib.orc().orc_prim_bbox(osc.h, p, a3.ctypes.data_as(fp), b3.ctypes.data_as(fp))
            orclib.ref().ref_prim_bbox(rs, p, c3.ctypes.data_as(fp), d3.ctypes.data_as(fp))
            assert bits(a3).tolist() == bits(c3).tolist() and bits(b3).tolist() == bits(d3).tolist(), f"bbox prim {p}"
            orclib.orc().orc_prim_center(osc.h, p, a3.ctypes.data_as(fp))
            orclib.ref().ref_prim_center(rs, p, c3.ctypes.data_as(fp))
            assert bits(a3).tolist() == bits(c3).tolist(), f"center prim {p}"
        # intersections and normals
        rays, _ = rays_for_scene(rng, sd, osc, 300)
        hits = 0
        for p in prims[:120]:
            p = int(p)
            for r in rays[:: max(1, len(rays) // 60)]:
                o, d = r[:3].copy(), r[3:].copy()
                t0 = orclib.orc().orc_prim_intersect(osc.h, p, o.ctypes.data_as(fp), d.ctypes.data_as(fp))
                t1 = orclib.ref().ref_prim_intersect(rs, p, o.ctypes.data_as(fp), d.ctypes.data_as(fp))
                assert bits([t0])[0] == bits([t1])[0], f"intersect prim {p}: {t0} vs {t1}"
                if t0 > 0:
                    hits += 1
                    orclib.orc().orc_prim_normal(osc.h, p, o.ctypes.data_as(fp), d.ctypes.data_as(fp), C.c_float(t0), a3.ctypes.data_as(fp))
                    orclib.ref().ref_prim_normal(rs, p, o.ctypes.data_as(fp), d.ctypes.data_as(fp), C.c_float(t1), c3.ctypes.data_as(fp))
                    assert bits(a3).tolist() == bits(c3).tolist(), f"normal prim {p}"
        # aimed rays: each ray against the primitive it was aimed at
        for i in range(min(osc.n_bvh_prims, 300)):
            p = int(rng.integers(0, osc.n_bvh_prims))
            orclib.orc().orc_prim_center(osc.h, p, c3.ctypes.data_as(fp))
            o = rng.normal(0, 5, 3).astype(np.float32)
            d = (c3 - o).astype(np.float32)
            t0 = orclib.orc().orc_prim_intersect(osc.h, p, o.ctypes.data_as(fp), d.ctypes.data_as(fp))
            t1 = orclib.ref().ref_prim_intersect(rs, p, o.ctypes.data_as(fp), d.ctypes.data_as(fp))
            assert bits([t0])[0] == bits([t1])[0]
            if t0 > 0:
                hits += 1
                orclib.orc().orc_prim_normal(osc.h, p, o.ctypes.data_as(fp), d.ctypes.data_as(fp), C.c_float(t0), a3.ctypes.data_as(fp))
                orclib.ref().ref_prim_normal(rs, p, o.ctypes.data_as(fp), d.ctypes.data_as(fp), C.c_float(t1), b3.ctypes.data_as(fp))
                assert bits(a3).tolist() == bits(b3).tolist()
        assert hits > 0
    finally:
        orclib.ref().ref_scene_destroy(rs)
        osc.close()


def test_box_intersect_bit_exact():
    rng = np.random.default_rng(7)
    n = 20000
    for i in range(n):
        mn = rng.normal(0, 3, 3).astype(np.float32)
        mx = (mn + np.abs(rng.normal(0, 2, 3))).astype(np.float32)
        o = rng.normal(0, 4, 3).astype(np.float32)
        d = rng.normal(0, 1, 3).astype(np.float32)
        if i % 5 == 0:
            d[int(rng.integers(0, 3))] = np.float32(rng.choice([0.0, 5e-5, -5e-5, 1e-4, -1e-4]))
        if i % 11 == 0:
            o = ((mn + mx) / 2).astype(np.float32)      # origin inside
        a = orclib.orc().orc_box_intersect(o.ctypes.data_as(fp), d.ctypes.data_as(fp), mn.ctypes.data_as(fp), mx.ctypes.data_as(fp))
        b = orclib.ref().ref_box_intersect(o.ctypes.data_as(fp), d.ctypes.data_as(fp), mn.ctypes.data_as(fp), mx.ctypes.data_as(fp))
        assert bits([a])[0] == bits([b])[0]


@pytest.mark.parametrize("shading", [abi.RT_PHONG, abi.RT_COOK_TORRANCE])
def test_shaders_bit_exact(shading):
    rng = np.random.default_rng(99)
    out0, out1 = np.zeros(3, np.float32), np.zeros(3, np.float32)

    def unit(v):
        v = v.astype(np.float64)
        return (v / np.linalg.norm(v)).astype(np.float32)

    for i in range(20000):
        clr = rng.random(3).astype(np.float32)
        lc = (rng.random(3) * 1.5).astype(np.float32)
        amb, dif, spec = (np.float32(x) for x in rng.random(3))
        rough = np.float32(rng.choice([0.05, 0.01, 0.1, 0.3, 1.0, 10.0]))
        eye, ld, nrm = unit(rng.normal(size=3)), unit(rng.normal(size=3)), unit(rng.normal(size=3))
        sh = int(i % 3 == 0)
        args = (clr.ctypes.data_as(fp), C.c_float(amb), C.c_float(dif), C.c_float(spec), C.c_float(rough),
                eye.ctypes.data_as(fp), ld.ctypes.data_as(fp), lc.ctypes.data_as(fp), nrm.ctypes.data_as(fp), sh)
        orclib.orc().orc_shade(shading, *args, out0.ctypes.data_as(fp))
        orclib.ref().ref_shade(shading, *args, out1.ctypes.data_as(fp))
        assert bits(out0).tolist() == bits(out1).tolist(), (i, out0, out1)


def test_texture_coordinates_bit_exact():
    """Triangle::UVAt (Triangle.h:44-52) on the textured mesh: the reference's own code against the oracle."""
    path = os.path.join(orclib.REFERENCE_DIR, "blitzcrank.pov")
    if not os.path.exists(path):
        pytest.skip("blitzcrank.pov not present")
    sd = orclib.ref_parse(path)
    assert [t["name"] for t in sd.textures] == ["blitz.bmp"]
    assert (sd.smooth_triangles["mod"]["mat"]["tex_id"] == 0).all()
    osc = orclib.OracleScene(sd)
    rs = orclib.ref().ref_scene_create(C.byref(osc.desc))
    try:
        rng = np.random.default_rng(8)
        c3, a2, b2 = np.zeros(3, np.float32), np.zeros(2, np.float32), np.zeros(2, np.float32)
        hits = 0
        for _ in range(2000):
            p = int(rng.integers(len(sd.spheres), osc.n_bvh_prims))
            orclib.orc().orc_prim_center(osc.h, p, c3.ctypes.data_as(fp))
            o = rng.normal(0, 5, 3).astype(np.float32)
            d = (c3 + rng.normal(0, 0.01, 3).astype(np.float32) - o).astype(np.float32)
            t = orclib.orc().orc_prim_intersect(osc.h, p, o.ctypes.data_as(fp), d.ctypes.data_as(fp))
            if t <= 0:
                continue
            hits += 1
            orclib.orc().orc_prim_uv(osc.h, p, o.ctypes.data_as(fp), d.ctypes.data_as(fp), C.c_float(t), a2.ctypes.data_as(fp))
            orclib.ref().ref_prim_uv(rs, p, o.ctypes.data_as(fp), d.ctypes.data_as(fp), C.c_float(t), b2.ctypes.data_as(fp))
            assert bits(a2).tolist() == bits(b2).tolist()
        assert hits > 500
    finally:
        orclib.ref().ref_scene_destroy(rs)
        osc.close()
