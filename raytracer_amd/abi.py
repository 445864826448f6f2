"""ctypes / numpy view of the flat C ABI declared in include/raytracer_amd.h.

Only layouts and the library loader live here; nothing in this module computes.
The loader fails loudly when the HIP library has not been built: there is no
CPU fallback for the render path.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RT_LIBRARY names another build of the same library (A/B measurements); it is still this project's HIP library
LIB_PATH = os.environ.get("RT_LIBRARY") or os.path.join(_HERE, "libraytracer_amd.so")

# status codes (include/raytracer_amd.h)
RT_OK = 0
RT_ERR_BAD_ARGUMENT = 1
RT_ERR_BAD_SAMPLE_COUNT = 2
RT_ERR_NO_DEVICE = 3
RT_ERR_HIP = 4
RT_ERR_UNSUPPORTED = 5
RT_ERR_PARSE = 6
RT_ERR_IO = 7
RT_ERR_NO_CAMERA = 8
RT_ERR_TOO_LARGE = 9

RT_PHONG, RT_COOK_TORRANCE = 0, 1
RT_JITTER_NONE, RT_JITTER_HASH = 0, 1
RT_OUT_FRAME, RT_OUT_TILES = 0, 1
RT_TILE = 32
RT_MAX_RECURSE = 6
RT_TRACE_RAYS_STATIC_MAX = 524288
RT_SCENE_DEFAULT, RT_SCENE_COUNTERS, RT_SCENE_FAST_BVH, RT_SCENE_GPU_BVH, RT_SCENE_NO_FILTER, RT_SCENE_FILTER_ALWAYS = 0, 1, 2, 4, 8, 16

f32, i32 = np.float32, np.int32

# numpy mirrors of the POD records (all members are 4-byte: no padding)
material_dtype = np.dtype([("clr", f32, 3), ("alpha", f32), ("amb", f32), ("dif", f32), ("spec", f32),
                           ("rough", f32), ("refl", f32), ("refr", f32), ("ior", f32), ("tex_id", i32)])
modifier_dtype = np.dtype([("mat", material_dtype), ("trans", f32, 16), ("inv_trans", f32, 16)])
sphere_dtype = np.dtype([("c", f32, 3), ("r", f32), ("mod", modifier_dtype)])
triangle_dtype = np.dtype([("p1", f32, 3), ("p2", f32, 3), ("p3", f32, 3),
                           ("vt1", f32, 2), ("vt2", f32, 2), ("vt3", f32, 2), ("mod", modifier_dtype)])
box_dtype = np.dtype([("p1", f32, 3), ("p2", f32, 3), ("mod", modifier_dtype)])
smooth_triangle_dtype = np.dtype([("p1", f32, 3), ("p2", f32, 3), ("p3", f32, 3),
                                  ("n1", f32, 3), ("n2", f32, 3), ("n3", f32, 3),
                                  ("vt1", f32, 2), ("vt2", f32, 2), ("vt3", f32, 2), ("mod", modifier_dtype)])
plane_dtype = np.dtype([("n", f32, 3), ("d", f32), ("mod", modifier_dtype)])
light_dtype = np.dtype([("pos", f32, 3), ("clr", f32, 3)])
bvh_node_dtype = np.dtype([("lmin", f32, 3), ("lmax", f32, 3), ("rmin", f32, 3), ("rmax", f32, 3),
                           ("left", i32), ("right", i32), ("pad", i32, 2)])
assert material_dtype.itemsize == 48 and modifier_dtype.itemsize == 176
assert sphere_dtype.itemsize == 192 and triangle_dtype.itemsize == 236 and box_dtype.itemsize == 200
assert smooth_triangle_dtype.itemsize == 272 and plane_dtype.itemsize == 192 and light_dtype.itemsize == 24
assert bvh_node_dtype.itemsize == 64


class Camera(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("up", C.c_float * 3), ("right", C.c_float * 3), ("look_at", C.c_float * 3)]


class Texture(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("rgba", C.c_void_p), ("name", C.c_char * 128)]


class SceneDesc(C.Structure):
    _fields_ = [("camera", Camera), ("has_camera", C.c_int32),
                ("n_spheres", C.c_int32), ("spheres", C.c_void_p),
                ("n_triangles", C.c_int32), ("triangles", C.c_void_p),
                ("n_boxes", C.c_int32), ("boxes", C.c_void_p),
                ("n_smooth_triangles", C.c_int32), ("smooth_triangles", C.c_void_p),
                ("n_planes", C.c_int32), ("planes", C.c_void_p),
                ("n_lights", C.c_int32), ("lights", C.c_void_p),
                ("n_textures", C.c_int32), ("textures", C.POINTER(Texture))]


class RenderParams(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("sample_count", C.c_int32),
                ("shading", C.c_int32), ("jitter", C.c_int32), ("seed", C.c_uint32),
                ("max_recurse", C.c_int32), ("tile_first", C.c_int32), ("tile_stride", C.c_int32),
                ("out_layout", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("rays_primary", C.c_uint64), ("rays_shadow", C.c_uint64), ("rays_reflect", C.c_uint64),
                ("rays_refract", C.c_uint64), ("rays_total", C.c_uint64), ("shadow_skipped", C.c_uint64),
                ("shade_nodes", C.c_uint64), ("samples", C.c_uint64),
                ("ms_total", C.c_float), ("ms_trace", C.c_float), ("ms_light", C.c_float),
                ("ms_combine", C.c_float), ("ms_resolve", C.c_float),
                ("wave_exit_mean_ms", C.c_float), ("wave_exit_last_ms", C.c_float),
                ("launches_trace", C.c_int32), ("attempts", C.c_int32),
                ("node_visits", C.c_uint64), ("prim_tests", C.c_uint64), ("tie_events", C.c_uint64),
                ("node_visits_shadow", C.c_uint64), ("prim_tests_shadow", C.c_uint64), ("tie_events_shadow", C.c_uint64),
                ("wave_node_steps", C.c_uint64), ("wave_leaf_steps", C.c_uint64),
                ("lane_steps_at_leaf", C.c_uint64), ("lane_steps_no_ray", C.c_uint64), ("wave_refills", C.c_uint64),
                ("wave_time_refill", C.c_float), ("wave_time_node", C.c_float), ("wave_time_leaf", C.c_float), ("second_tree", C.c_int32),
                ("max_ray_steps", C.c_uint64), ("max_spawn_list", C.c_uint64),
                ("hits_certified", C.c_uint64), ("hits_contested", C.c_uint64), ("hits_unreached", C.c_uint64),
                ("shadows_certified", C.c_uint64), ("primaries_culled", C.c_uint64), ("shadows_unneeded", C.c_uint64), ("hits_guided", C.c_uint64),
                ("wave_steps_both_trees", C.c_uint64), ("wave_steps_second_tree_only", C.c_uint64)]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


class SceneInfo(C.Structure):
    _fields_ = [("n_bvh_prims", C.c_int32), ("n_planes", C.c_int32), ("n_lights", C.c_int32),
                ("n_nodes", C.c_int32), ("n_materials", C.c_int32), ("tree_depth", C.c_int32),
                ("root", C.c_int32), ("device_bytes", C.c_uint64),
                ("ms_build_host", C.c_float), ("ms_upload", C.c_float),
                ("ms_build_device", C.c_float), ("filter_depth", C.c_int32)]


# every symbol include/raytracer_amd.h and include/raytracer_tokens.h declare
EXPORTED_SYMBOLS = [
    "rt_device_count", "rt_last_error", "rt_scene_create", "rt_scene_clone", "rt_scene_create_host", "rt_scene_destroy", "rt_scene_get_info",
    "rt_scene_get_bvh", "rt_scene_set_camera", "rt_render", "rt_render_device", "rt_tile_count", "rt_tiles_live",
    "rt_sequence_create", "rt_sequence_destroy", "rt_sequence_frames_in_flight", "rt_sequence_submit", "rt_sequence_wait",
    "rt_multi_create", "rt_multi_create_pipelined", "rt_multi_frames_in_flight", "rt_multi_submit", "rt_multi_wait", "rt_multi_destroy", "rt_multi_device_count", "rt_multi_set_camera", "rt_multi_render", "rt_multi_render_device",
    "rt_untile_device", "rt_trace_primary", "rt_frame_primary_hits", "rt_trace_rays", "rt_trace_rays_static", "rt_render_samples",
    "rt_parse_pov", "rt_scene_desc_free", "rt_load_bmp", "rt_write_tga", "launch_kernel",
]

_lib = None


def load_library(path=None):
    """Load libraryrtracer_amd.so (built in-tree by __graft_entry__.build()).  Raises if absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError(
            f"{p} is missing: build the HIP library first (python -c 'import __graft_entry__ as g; g.build()'). "
            "The render path has no CPU fallback.")
    lib = C.CDLL(p, mode=C.RTLD_GLOBAL)
    vp, i32c, i64c = C.c_void_p, C.c_int32, C.c_int64
    lib.rt_device_count.restype = C.c_int
    lib.rt_last_error.restype = C.c_char_p
    lib.rt_scene_create.argtypes = [C.POINTER(SceneDesc), C.c_int, C.c_int, C.POINTER(vp)]
    lib.rt_scene_create_host.argtypes = [C.POINTER(SceneDesc), C.POINTER(vp)]
    lib.rt_scene_clone.argtypes = [vp, C.c_int, C.POINTER(vp)]
    lib.rt_scene_destroy.argtypes = [vp]
    lib.rt_scene_destroy.restype = None
    lib.rt_scene_get_info.argtypes = [vp, C.POINTER(SceneInfo)]
    lib.rt_scene_get_bvh.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    lib.rt_scene_set_camera.argtypes = [vp, C.POINTER(Camera)]
    lib.rt_render.argtypes = [vp, C.POINTER(RenderParams), vp, C.POINTER(Stats)]
    lib.rt_render_device.argtypes = [vp, C.POINTER(RenderParams), vp, vp, C.POINTER(Stats)]
    lib.rt_tile_count.argtypes = [C.c_int] * 4
    lib.rt_tiles_live.argtypes = [vp, C.POINTER(RenderParams), vp, C.c_int]
    lib.rt_untile_device.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, C.c_int, vp, vp]
    lib.rt_trace_primary.argtypes = [vp, C.POINTER(RenderParams), vp, vp]
    lib.rt_frame_primary_hits.argtypes = [vp, vp, vp, i64c]
    lib.rt_trace_rays.argtypes = [vp, i64c, vp, vp, vp, vp]
    lib.rt_trace_rays_static.argtypes = [vp, i64c, vp, vp, vp, vp]
    lib.rt_sequence_create.argtypes = [C.POINTER(SceneDesc), C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    lib.rt_sequence_destroy.argtypes = [vp]
    lib.rt_sequence_destroy.restype = None
    lib.rt_sequence_frames_in_flight.argtypes = [vp]
    lib.rt_sequence_submit.argtypes = [vp, C.POINTER(RenderParams), C.POINTER(Camera), vp]
    lib.rt_sequence_wait.argtypes = [vp, C.POINTER(Stats)]
    lib.rt_multi_create.argtypes = [C.POINTER(SceneDesc), C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(vp)]
    lib.rt_multi_create_pipelined.argtypes = [C.POINTER(SceneDesc), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    lib.rt_multi_frames_in_flight.argtypes = [vp]
    lib.rt_multi_submit.argtypes = [vp, C.POINTER(RenderParams), C.POINTER(Camera), vp]
    lib.rt_multi_wait.argtypes = [vp, C.POINTER(Stats)]
    lib.rt_multi_destroy.argtypes = [vp]
    lib.rt_multi_destroy.restype = None
    lib.rt_multi_device_count.argtypes = [vp]
    lib.rt_multi_set_camera.argtypes = [vp, C.POINTER(Camera)]
    lib.rt_multi_render.argtypes = [vp, C.POINTER(RenderParams), vp, C.POINTER(Stats)]
    lib.rt_multi_render_device.argtypes = [vp, C.POINTER(RenderParams), vp, C.POINTER(Stats)]
    lib.rt_render_samples.argtypes = [vp, C.POINTER(RenderParams), vp, C.POINTER(Stats)]
    lib.rt_parse_pov.argtypes = [C.c_char_p, C.POINTER(C.POINTER(SceneDesc))]
    lib.rt_scene_desc_free.argtypes = [C.POINTER(SceneDesc)]
    lib.rt_scene_desc_free.restype = None
    lib.rt_write_tga.argtypes = [C.c_char_p, C.c_int, C.c_int, vp]
    lib.rt_load_bmp.argtypes = [C.c_char_p, C.POINTER(Texture)]
    if path is None:
        _lib = lib
    return lib
