"""ctypes bindings of the test oracle (oracle/liborc.so) and, when it has been
built in this container, of the reference's own classes (oracle/_ref/libref.so).

Test infrastructure: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg only.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from raytracer_amd import abi
from raytracer_amd.scene import SceneData

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC_DIR = os.path.join(ROOT, "oracle")
ORC_PATH = os.path.join(ORC_DIR, "liborc.so")
REF_PATH = os.path.join(ORC_DIR, "_ref", "libref.so")
REFERENCE_DIR = "/root/reference"


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "rays_primary", "rays_shadow", "rays_reflect", "rays_refract", "shadow_selfmiss", "shade_nodes",
        "box_tests", "node_first_visits", "node_revisits", "prim_tests", "plane_tests", "tie_events")] + [
        ("max_stack", C.c_int32)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class OrcParams(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("sample_count", C.c_int32),
                ("shading", C.c_int32), ("jitter", C.c_int32), ("seed", C.c_uint32),
                ("max_recurse", C.c_int32), ("threads", C.c_int32)]


_orc = None
_ref = None
fp = C.POINTER(C.c_float)


def _f3(a):
    return np.ascontiguousarray(a, dtype=np.float32).ctypes.data_as(fp)


def build_oracle():
    subprocess.run(["make", "-C", ORC_DIR, "liborc.so"], check=True, capture_output=True)


def orc():
    global _orc
    if _orc is None:
        if not os.path.exists(ORC_PATH):
            build_oracle()
        lib = C.CDLL(ORC_PATH)
        vp = C.c_void_p
        lib.orc_scene_create.restype = vp
        lib.orc_scene_create.argtypes = [C.POINTER(abi.SceneDesc)]
        lib.orc_scene_destroy.argtypes = [vp]
        lib.orc_n_bvh_prims.argtypes = [vp]
        lib.orc_n_planes.argtypes = [vp]
        lib.orc_leaf_order.argtypes = [vp]
        lib.orc_leaf_order.restype = C.POINTER(C.c_int32)
        lib.orc_bvh_trace.argtypes = [vp, C.c_char_p, C.c_size_t]
        lib.orc_bvh_trace.restype = C.c_size_t
        lib.orc_flatten_bvh.argtypes = [vp, vp, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        lib.orc_closest.argtypes = [vp, fp, fp, C.c_float, C.POINTER(C.c_int32), fp, C.POINTER(Counters)]
        lib.orc_prim_intersect.argtypes = [vp, C.c_int32, fp, fp]
        lib.orc_prim_intersect.restype = C.c_float
        lib.orc_prim_normal.argtypes = [vp, C.c_int32, fp, fp, C.c_float, fp]
        lib.orc_prim_uv.argtypes = [vp, C.c_int32, fp, fp, C.c_float, fp]
        lib.orc_texture_sample.argtypes = [C.POINTER(abi.Texture), C.c_float, C.c_float, fp]
        lib.orc_prim_bbox.argtypes = [vp, C.c_int32, fp, fp]
        lib.orc_prim_center.argtypes = [vp, C.c_int32, fp]
        lib.orc_box_intersect.argtypes = [fp, fp, fp, fp]
        lib.orc_box_intersect.restype = C.c_float
        lib.orc_shade.argtypes = [C.c_int, fp, C.c_float, C.c_float, C.c_float, C.c_float, fp, fp, fp, fp, C.c_int, fp]
        lib.orc_camera_ray.argtypes = [vp, C.POINTER(OrcParams), C.c_int, C.c_int, fp, fp]
        lib.orc_render.argtypes = [vp, C.POINTER(OrcParams), vp, vp, vp, vp, C.POINTER(Counters)]
        lib.orc_jitter_uniforms.argtypes = [C.c_uint32, C.c_uint32, fp, fp]
        lib.orc_set_sphere_sqrt_double.argtypes = [C.c_int]
        lib.orc_interior_mask.argtypes = [vp, C.POINTER(OrcParams), vp]
        lib.orc_brute_primary.argtypes = [vp, C.POINTER(OrcParams), vp, vp]
        _orc = lib
    return _orc


def have_ref():
    return os.path.exists(REF_PATH)


def ref():
    global _ref
    if _ref is None:
        lib = C.CDLL(REF_PATH)
        vp = C.c_void_p
        lib.ref_parse_pov.argtypes = [C.c_char_p, C.POINTER(C.POINTER(abi.SceneDesc))]
        lib.ref_scene_desc_free.argtypes = [C.POINTER(abi.SceneDesc)]
        lib.ref_scene_create.restype = vp
        lib.ref_scene_create.argtypes = [C.POINTER(abi.SceneDesc)]
        lib.ref_scene_destroy.argtypes = [vp]
        lib.ref_prim_intersect.argtypes = [vp, C.c_int, fp, fp]
        lib.ref_prim_intersect.restype = C.c_float
        lib.ref_prim_normal.argtypes = [vp, C.c_int, fp, fp, C.c_float, fp]
        lib.ref_prim_uv.argtypes = [vp, C.c_int, fp, fp, C.c_float, fp]
        lib.ref_prim_bbox.argtypes = [vp, C.c_int, fp, fp]
        lib.ref_prim_center.argtypes = [vp, C.c_int, fp]
        lib.ref_box_intersect.argtypes = [fp, fp, fp, fp]
        lib.ref_box_intersect.restype = C.c_float
        lib.ref_combine_bbox.argtypes = [fp] * 6
        lib.ref_shade.argtypes = [C.c_int, fp, C.c_float, C.c_float, C.c_float, C.c_float, fp, fp, fp, fp, C.c_int, fp]
        lib.ref_light_dir.argtypes = [vp, C.c_int, fp, fp]
        lib.ref_light_feeler.argtypes = [vp, C.c_int, fp, fp, fp]
        lib.ref_write_tga.argtypes = [C.c_char_p, C.c_int, C.c_int, vp]
        _ref = lib
    return _ref


def ref_parse(path):
    """Parse with the reference's own POVRayParser; returns SceneData."""
    out = C.POINTER(abi.SceneDesc)()
    st = ref().ref_parse_pov(path.encode(), C.byref(out))
    if st != 0:
        raise RuntimeError(f"reference parser status {st} on {path}")
    try:
        return SceneData.from_desc(out.contents)
    finally:
        ref().ref_scene_desc_free(out)


class OracleScene:
    """orc_scene handle over a SceneData."""

    def __init__(self, scene: SceneData):
        self.scene = scene
        self.desc, self._keep = scene.to_desc()
        self.h = orc().orc_scene_create(C.byref(self.desc))
        self.n_bvh_prims = orc().orc_n_bvh_prims(self.h)
        self.n_planes = orc().orc_n_planes(self.h)

    def close(self):
        if self.h:
            orc().orc_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def leaf_order(self):
        p = orc().orc_leaf_order(self.h)
        return np.ctypeslib.as_array(p, shape=(max(self.n_bvh_prims, 1),))[:self.n_bvh_prims].copy()

    def bvh_trace(self):
        n = orc().orc_bvh_trace(self.h, None, 0)
        buf = C.create_string_buffer(n + 1)
        orc().orc_bvh_trace(self.h, buf, n + 1)
        return buf.value.decode()

    def flatten(self):
        cap = max(self.n_bvh_prims, 1)
        nodes = np.zeros(cap, dtype=abi.bvh_node_dtype)
        root, depth = C.c_int32(), C.c_int32()
        n = orc().orc_flatten_bvh(self.h, nodes.ctypes.data, cap, C.byref(root), C.byref(depth))
        return nodes[:n].copy(), root.value, depth.value

    def closest(self, o, d, to_beat=-3.4028234663852886e38, counters=None):
        prim, t = C.c_int32(), C.c_float()
        orc().orc_closest(self.h, _f3(o), _f3(d), C.c_float(to_beat), C.byref(prim), C.byref(t),
                          C.byref(counters) if counters is not None else None)
        return prim.value, t.value

    def closest_batch(self, rays, to_beat=None):
        rays = np.ascontiguousarray(rays, dtype=np.float32)
        n = rays.shape[0]
        prims = np.empty(n, np.int32)
        ts = np.empty(n, np.float32)
        for i in range(n):
            tb = -3.4028234663852886e38 if to_beat is None else float(to_beat[i])
            prims[i], ts[i] = self.closest(rays[i, 0:3], rays[i, 3:6], tb)
        return prims, ts

    def brute_primary(self, width, height, sample_count=1, jitter=abi.RT_JITTER_NONE, seed=0, threads=0):
        """(prim, t) of every primary ray with no tree: every primitive and plane tested, the nearest kept."""
        s = int(round(sample_count ** 0.5))
        n = width * s * height * s
        p = OrcParams(width, height, sample_count, abi.RT_PHONG, jitter, seed, 0, threads)
        prim, t = np.full(n, -2, np.int32), np.zeros(n, np.float32)
        orc().orc_brute_primary(self.h, C.byref(p), prim.ctypes.data, t.ctypes.data)
        return prim, t

    def interior_mask(self, width, height, sample_count=1, shading=abi.RT_PHONG, max_recurse=0, threads=0):
        """(height, width) bool: pixels whose shade trees make the same decisions wherever the jitter puts their samples."""
        p = OrcParams(width, height, sample_count, shading, abi.RT_JITTER_NONE, 0, max_recurse, threads)
        mask = np.zeros((height, width), np.uint8)
        orc().orc_interior_mask(self.h, C.byref(p), mask.ctypes.data)
        return mask.astype(bool)

    def render(self, width, height, sample_count=1, shading=abi.RT_PHONG, jitter=abi.RT_JITTER_NONE, seed=0,
               max_recurse=0, threads=0, want_samples=True, want_primary=True):
        s = int(round(sample_count ** 0.5))
        ns = width * s * height * s
        p = OrcParams(width, height, sample_count, shading, jitter, seed, max_recurse, threads)
        rgb = np.zeros((height * s, width * s, 3), np.float32) if want_samples else None
        rgba = np.zeros((height, width, 4), np.uint8)
        prim = np.full(ns, -2, np.int32) if want_primary else None
        t = np.zeros(ns, np.float32) if want_primary else None
        cn = Counters()
        orc().orc_render(self.h, C.byref(p), rgb.ctypes.data if rgb is not None else None, rgba.ctypes.data,
                         prim.ctypes.data if prim is not None else None, t.ctypes.data if t is not None else None,
                         C.byref(cn))
        return {"rgb": rgb, "rgba": rgba, "prim": prim, "t": t, "counters": cn.as_dict()}
