"""Host-side mirror of the reference's operator interface for the render path.

`launch_kernel(scene, stype, width, height, output, sample_count)` has the
argument order and meaning of the reference's entry point (kernel.h:30-31);
`Scene` is the handle-based form of the same call for callers that render more
than one frame of a scene (the reference rebuilds everything per call).

Everything here goes through the C ABI of libraytracer_amd.so (HIP, gfx950).
There is no CPU path: without the library or without a GPU these calls raise.
"""
import ctypes as C

import numpy as np

from . import abi
from .scene import SceneData

PHONG, COOK_TORRANCE = abi.RT_PHONG, abi.RT_COOK_TORRANCE


class RaytracerError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"raytracer_amd error {code}: {text}")
        self.code = code


def _check(lib, rc):
    if rc != abi.RT_OK:
        raise RaytracerError(rc, (lib.rt_last_error() or b"").decode(errors="replace"))


def device_count():
    return abi.load_library().rt_device_count()


def parse_pov(path):
    """POVRayParser::parseFile (POVRayParser.cpp:14-62) -> SceneData."""
    lib = abi.load_library()
    out = C.POINTER(abi.SceneDesc)()
    _check(lib, lib.rt_parse_pov(str(path).encode(), C.byref(out)))
    try:
        return SceneData.from_desc(out.contents)
    finally:
        lib.rt_scene_desc_free(out)


def load_bmp(path):
    """readBMP (kernel.cu:35-61) -> (H, W, 4) uint8 texels, row 0 = bottom row of the picture."""
    lib = abi.load_library()
    t = abi.Texture()
    _check(lib, lib.rt_load_bmp(str(path).encode(), C.byref(t)))
    try:
        buf = (C.c_char * (t.width * t.height * 4)).from_address(t.rgba)
        return np.frombuffer(buf, dtype=np.uint8).reshape(t.height, t.width, 4).copy()
    finally:
        C.CDLL(None).free(C.c_void_p(t.rgba))


def write_tga(path, rgba):
    """Frame (H, W, 4) uint8, row 0 = bottom row -> TGA as the reference's driver writes it."""
    lib = abi.load_library()
    rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
    h, w = rgba.shape[:2]
    _check(lib, lib.rt_write_tga(str(path).encode(), w, h, rgba.ctypes.data))


def make_params(width, height, sample_count=1, shading=PHONG, jitter=abi.RT_JITTER_HASH, seed=0, max_recurse=0,
                tile_first=0, tile_stride=1, out_layout=abi.RT_OUT_FRAME):
    return abi.RenderParams(width, height, sample_count, shading, jitter, seed, max_recurse, tile_first, tile_stride,
                            out_layout)


class Scene:
    """Device-resident scene: objects in the reference's creation order, its BVH, flattened (rt_scene_create)."""

    def __init__(self, data: SceneData, device=0, counters=False, host_only=False, fast_bvh=False, gpu_bvh=False, miss_filter=None):
        self._lib = abi.load_library()
        self.data = data
        desc, keep = data.to_desc()
        h = C.c_void_p()
        if host_only:
            _check(self._lib, self._lib.rt_scene_create_host(C.byref(desc), C.byref(h)))
        else:
            _check(self._lib, self._lib.rt_scene_create(C.byref(desc), device,
                                                        (abi.RT_SCENE_COUNTERS if counters else 0) | (abi.RT_SCENE_FAST_BVH if fast_bvh else 0)
                                                        | (abi.RT_SCENE_GPU_BVH if gpu_bvh else 0) | (0 if miss_filter is None else abi.RT_SCENE_FILTER_ALWAYS if miss_filter else abi.RT_SCENE_NO_FILTER),
                                                        C.byref(h)))
        self._h = h
        self.device = device
        del keep

    def clone(self, device=None):
        """A replica on `device` (default: this scene's) that shares this scene's host side (rt_scene_clone)."""
        h = C.c_void_p()
        _check(self._lib, self._lib.rt_scene_clone(self._h, self.device if device is None else device, C.byref(h)))
        other = Scene.__new__(Scene)
        other._lib, other.data, other._h, other.device = self._lib, self.data, h, self.device if device is None else device
        return other

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rt_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---------------------------------------------------------------- queries
    def info(self):
        i = abi.SceneInfo()
        _check(self._lib, self._lib.rt_scene_get_info(self._h, C.byref(i)))
        return {n: getattr(i, n) for n, _ in i._fields_}

    def bvh(self):
        """(nodes[n_nodes] structured array, leaf_order[n_bvh_prims]) -- host copies."""
        info = self.info()
        pn, po = C.c_void_p(), C.c_void_p()
        _check(self._lib, self._lib.rt_scene_get_bvh(self._h, C.byref(pn), C.byref(po)))
        nn, npr = info["n_nodes"], info["n_bvh_prims"]
        nodes = np.zeros(nn, dtype=abi.bvh_node_dtype)
        if nn:
            C.memmove(nodes.ctypes.data, pn.value, nn * 64)
        order = np.zeros(npr, dtype=np.int32)
        if npr:
            C.memmove(order.ctypes.data, po.value, npr * 4)
        return nodes, order

    def set_camera(self, pos, up, right, look_at):
        cam = abi.Camera()
        for k, v in (("pos", pos), ("up", up), ("right", right), ("look_at", look_at)):
            getattr(cam, k)[:] = [float(x) for x in v]
        _check(self._lib, self._lib.rt_scene_set_camera(self._h, C.byref(cam)))

    def tiles_live(self, width, height, sample_count=1, tile_first=0, tile_stride=1):
        """Per selected tile: 1 if a frame of this size traces it, 0 if it is left out of the launch (no ray through
        it can reach the scene's bounding box).  A host computation; works on host-only scenes."""
        p = make_params(width, height, sample_count, PHONG, abi.RT_JITTER_HASH, 0, 0, tile_first, tile_stride)
        n = self._lib.rt_tile_count(width, height, tile_first, tile_stride)
        live = np.zeros(max(n, 1), np.uint8)
        _check(self._lib, self._lib.rt_tiles_live(self._h, C.byref(p), live.ctypes.data, n))
        return live[:n]

    # ----------------------------------------------------------------- render
    def render(self, width, height, sample_count=1, shading=PHONG, jitter=abi.RT_JITTER_HASH, seed=0, max_recurse=0,
               tile_first=0, tile_stride=1):
        """Whole frame (or a tile subset) -> (H, W, 4) uint8 host array (row 0 = bottom), stats dict."""
        p = make_params(width, height, sample_count, shading, jitter, seed, max_recurse, tile_first, tile_stride)
        out = np.zeros((height, width, 4), np.uint8)
        st = abi.Stats()
        _check(self._lib, self._lib.rt_render(self._h, C.byref(p), out.ctypes.data, C.byref(st)))
        return out, st.as_dict()

    def render_device(self, params, out_ptr, stream_ptr=None):
        """Output stays in device memory at `out_ptr` (e.g. a torch tensor's data_ptr()); returns stats dict."""
        st = abi.Stats()
        _check(self._lib, self._lib.rt_render_device(self._h, C.byref(params), C.c_void_p(out_ptr),
                                                      C.c_void_p(stream_ptr) if stream_ptr else None, C.byref(st)))
        return st.as_dict()

    def trace_primary(self, width, height, sample_count=1, jitter=abi.RT_JITTER_NONE, seed=0):
        s = int(round(sample_count ** 0.5))
        n = width * s * height * s
        p = make_params(width, height, sample_count, PHONG, jitter, seed)
        prim = np.empty(n, np.int32)
        t = np.empty(n, np.float32)
        _check(self._lib, self._lib.rt_trace_primary(self._h, C.byref(p), prim.ctypes.data, t.ctypes.data))
        return prim, t

    def frame_primary_hits(self, width, height, sample_count=1):
        """(prim, t) of the primaries of the frame rendered last, as that frame's own launch stored them."""
        s = int(round(sample_count ** 0.5))
        n = width * s * height * s
        prim = np.empty(n, np.int32)
        t = np.empty(n, np.float32)
        _check(self._lib, self._lib.rt_frame_primary_hits(self._h, prim.ctypes.data, t.ctypes.data, n))
        return prim, t

    def trace_rays(self, rays, to_beat=None, static=False):
        """Closest hits of arbitrary rays (n x {o, d}); `static` goes through rt_trace_rays_static, the variant
        that allocates no device memory (at most abi.RT_TRACE_RAYS_STATIC_MAX rays)."""
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
        n = rays.shape[0]
        prim = np.empty(n, np.int32)
        t = np.empty(n, np.float32)
        tb = None if to_beat is None else np.ascontiguousarray(to_beat, dtype=np.float32)
        fn = self._lib.rt_trace_rays_static if static else self._lib.rt_trace_rays
        _check(self._lib, fn(self._h, n, rays.ctypes.data, tb.ctypes.data if tb is not None else None, prim.ctypes.data, t.ctypes.data))
        return prim, t

    def render_samples(self, width, height, sample_count=1, shading=PHONG, jitter=abi.RT_JITTER_NONE, seed=0,
                       max_recurse=0):
        s = int(round(sample_count ** 0.5))
        p = make_params(width, height, sample_count, shading, jitter, seed, max_recurse)
        rgb = np.zeros((height * s, width * s, 3), np.float32)
        st = abi.Stats()
        _check(self._lib, self._lib.rt_render_samples(self._h, C.byref(p), rgb.ctypes.data, C.byref(st)))
        return rgb, st.as_dict()


class Sequence:
    """A sequence of frames of one scene with several frames in flight on one device (rt_sequence_*): a scene replica,
    a stream and a host thread per frame in flight; submit() returns at once, wait() returns the oldest frame's stats."""

    def __init__(self, data: SceneData, device=0, frames_in_flight=2, counters=False, fast_bvh=False):
        self._lib = abi.load_library()
        desc, keep = data.to_desc()
        h = C.c_void_p()
        flags = (abi.RT_SCENE_COUNTERS if counters else 0) | (abi.RT_SCENE_FAST_BVH if fast_bvh else 0)
        _check(self._lib, self._lib.rt_sequence_create(C.byref(desc), device, flags, frames_in_flight, C.byref(h)))
        self._h = h
        del keep

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rt_sequence_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def frames_in_flight(self):
        return self._lib.rt_sequence_frames_in_flight(self._h)

    def submit(self, params, out_ptr, camera=None):
        """Queue a frame into device memory at `out_ptr`; camera = None or a dict of pos / up / right / look_at."""
        cam = None
        if camera is not None:
            cam = abi.Camera()
            for k in ("pos", "up", "right", "look_at"):
                getattr(cam, k)[:] = [float(x) for x in camera[k]]
        _check(self._lib, self._lib.rt_sequence_submit(self._h, C.byref(params), C.byref(cam) if cam is not None else None, C.c_void_p(out_ptr)))

    def wait(self):
        st = abi.Stats()
        _check(self._lib, self._lib.rt_sequence_wait(self._h, C.byref(st)))
        return st.as_dict()


class MultiScene:
    """One frame on several devices (rt_multi_*): a scene replica per entry of `devices`, tiles dealt round-robin,
    compact tile buffers gathered on the first device, un-tiled there.  A device may be listed more than once."""

    def __init__(self, data: SceneData, devices, counters=False, fast_bvh=False, frames_in_flight=1):
        self._lib = abi.load_library()
        desc, keep = data.to_desc()
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        h = C.c_void_p()
        flags = (abi.RT_SCENE_COUNTERS if counters else 0) | (abi.RT_SCENE_FAST_BVH if fast_bvh else 0)
        _check(self._lib, self._lib.rt_multi_create_pipelined(C.byref(desc), devs, len(devices), flags, frames_in_flight, C.byref(h)))
        self._h = h
        del keep

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rt_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def device_count(self):
        return self._lib.rt_multi_device_count(self._h)

    def set_camera(self, pos, up, right, look_at):
        cam = abi.Camera()
        for k, v in (("pos", pos), ("up", up), ("right", right), ("look_at", look_at)):
            getattr(cam, k)[:] = [float(x) for x in v]
        _check(self._lib, self._lib.rt_multi_set_camera(self._h, C.byref(cam)))

    def render(self, width, height, sample_count=1, shading=PHONG, jitter=abi.RT_JITTER_HASH, seed=0, max_recurse=0):
        p = make_params(width, height, sample_count, shading, jitter, seed, max_recurse)
        out = np.zeros((height, width, 4), np.uint8)
        st = abi.Stats()
        _check(self._lib, self._lib.rt_multi_render(self._h, C.byref(p), out.ctypes.data, C.byref(st)))
        return out, st.as_dict()

    def render_device(self, params, out_ptr):
        st = abi.Stats()
        _check(self._lib, self._lib.rt_multi_render_device(self._h, C.byref(params), C.c_void_p(out_ptr), C.byref(st)))
        return st.as_dict()

    def frames_in_flight(self):
        return self._lib.rt_multi_frames_in_flight(self._h)

    def submit(self, params, out_ptr, camera=None):
        """Queue a frame into memory of the first device at `out_ptr`; camera = None or a dict of pos / up / right / look_at."""
        cam = None
        if camera is not None:
            cam = abi.Camera()
            for k in ("pos", "up", "right", "look_at"):
                getattr(cam, k)[:] = [float(x) for x in camera[k]]
        _check(self._lib, self._lib.rt_multi_submit(self._h, C.byref(params), C.byref(cam) if cam is not None else None, C.c_void_p(out_ptr)))

    def wait(self):
        st = abi.Stats()
        _check(self._lib, self._lib.rt_multi_wait(self._h, C.byref(st)))
        return st.as_dict()


def tile_count(width, height, tile_first=0, tile_stride=1):
    return abi.load_library().rt_tile_count(width, height, tile_first, tile_stride)


def launch_kernel(data: SceneData, stype, width, height, output, sample_count, jitter=abi.RT_JITTER_HASH, device=0):
    """The reference's entry point (kernel.h:30-31) over a SceneData.

    `output` is a caller-owned uint8 array of width*height*4 bytes that is filled in place.  A sample count
    without an integer square root prints the reference's message and returns with `output` untouched
    (kernel.cu:707-712); any other failure raises.
    """
    s = int(sample_count ** 0.5)
    if s * s != sample_count:
        print(f"Invalid sample count: {sample_count}. Sample count for anti aliasing must have an integer square root",
              end="")
        return
    if data.camera is None:
        raise RaytracerError(abi.RT_ERR_NO_CAMERA, "scene has no camera")
    out = np.asarray(output)
    if out.dtype != np.uint8 or out.size != width * height * 4 or not out.flags["C_CONTIGUOUS"]:
        raise ValueError("output must be a contiguous uint8 array of width*height*4 bytes")
    with Scene(data, device=device) as sc:
        frame, _ = sc.render(width, height, sample_count, stype, jitter)
    out.reshape(-1)[:] = frame.reshape(-1)
