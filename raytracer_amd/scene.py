"""Host-side scene description: the numpy mirror of TKSceneData (TokenData.h:96-110).

A SceneData owns one structured numpy array per primitive class, in the order
the reference's parser would have pushed them, and converts to the POD
`rt_scene_desc` the C ABI takes.  No rendering logic here.
"""
import ctypes as C

import numpy as np

from . import abi

IDENTITY = np.eye(4, dtype=np.float32).reshape(16)

_ARRAYS = [("spheres", abi.sphere_dtype), ("triangles", abi.triangle_dtype), ("boxes", abi.box_dtype),
           ("smooth_triangles", abi.smooth_triangle_dtype), ("planes", abi.plane_dtype), ("lights", abi.light_dtype)]


def default_material(n=None):
    """Finish defaults of TKFinish (TokenData.h:36-38) and TKPigment (TokenData.h:24-25)."""
    m = np.zeros(() if n is None else n, dtype=abi.material_dtype)
    m["amb"], m["dif"], m["spec"], m["rough"] = 0.1, 0.6, 0.0, 0.05
    m["refl"], m["refr"], m["ior"], m["tex_id"] = 0.0, 0.0, 1.0, -1
    return m


def identity_modifier(n=None):
    mod = np.zeros(() if n is None else n, dtype=abi.modifier_dtype)
    mod["mat"] = default_material(n)
    mod["trans"] = IDENTITY
    mod["inv_trans"] = IDENTITY
    return mod


class SceneData:
    def __init__(self):
        self.camera = None  # dict(pos, up, right, look_at) of 3-float sequences
        for name, dt in _ARRAYS:
            setattr(self, name, np.zeros(0, dtype=dt))
        # image_map textures, indexed by material tex_id: {"name": str, "rgba": (H, W, 4) uint8 or None}; row 0 is
        # the bottom row of the picture (the order a .bmp stores it)
        self.textures = []

    # ------------------------------------------------------------------ C view
    def to_desc(self):
        """Returns (SceneDesc, keepalive) -- keep `keepalive` referenced while the desc is in use."""
        d = abi.SceneDesc()
        keep = []
        if self.camera is not None:
            d.has_camera = 1
            for k in ("pos", "up", "right", "look_at"):
                v = np.asarray(self.camera[k], dtype=np.float32)
                getattr(d.camera, k)[:] = [float(x) for x in v]
        for name, dt in _ARRAYS:
            arr = np.ascontiguousarray(getattr(self, name), dtype=dt)
            keep.append(arr)
            setattr(d, "n_" + name, int(arr.shape[0]))
            setattr(d, name, arr.ctypes.data if arr.shape[0] else None)
        if self.textures:
            tex = (abi.Texture * len(self.textures))()
            for i, t in enumerate(self.textures):
                tex[i].name = t["name"].encode()[:127]
                if t.get("rgba") is not None:
                    px = np.ascontiguousarray(t["rgba"], dtype=np.uint8)
                    keep.append(px)
                    tex[i].height, tex[i].width = px.shape[0], px.shape[1]
                    tex[i].rgba = px.ctypes.data
            keep.append(tex)
            d.n_textures = len(self.textures)
            d.textures = C.cast(tex, C.POINTER(abi.Texture))
        return d, keep

    @classmethod
    def from_desc(cls, desc):
        """Deep copy out of a C rt_scene_desc (e.g. one returned by rt_parse_pov)."""
        s = cls()
        if desc.has_camera:
            s.camera = {k: np.array(list(getattr(desc.camera, k)), dtype=np.float32)
                        for k in ("pos", "up", "right", "look_at")}
        for name, dt in _ARRAYS:
            n = getattr(desc, "n_" + name)
            ptr = getattr(desc, name)
            if n and ptr:
                buf = (C.c_char * (n * dt.itemsize)).from_address(ptr)
                setattr(s, name, np.frombuffer(buf, dtype=dt, count=n).copy())
        for i in range(desc.n_textures):
            t = desc.textures[i]
            px = None
            if t.rgba:
                buf = (C.c_char * (t.width * t.height * 4)).from_address(t.rgba)
                px = np.frombuffer(buf, dtype=np.uint8).reshape(t.height, t.width, 4).copy()
            s.textures.append({"name": t.name.decode(errors="replace"), "rgba": px})
        return s

    # ------------------------------------------------------------ persistence
    def save(self, path):
        cam = np.zeros(12, dtype=np.float32)
        if self.camera is not None:
            cam = np.concatenate([np.asarray(self.camera[k], dtype=np.float32)
                                  for k in ("pos", "up", "right", "look_at")])
        extra = {"texture_names": np.array([t["name"] for t in self.textures], dtype="U128")}
        for i, t in enumerate(self.textures):
            if t.get("rgba") is not None:
                extra[f"texture_{i}"] = np.ascontiguousarray(t["rgba"], dtype=np.uint8)
        np.savez_compressed(path, camera=cam, has_camera=np.int32(self.camera is not None),
                            **{name: getattr(self, name) for name, _ in _ARRAYS}, **extra)

    @classmethod
    def load(cls, path):
        z = np.load(path, allow_pickle=False)
        s = cls()
        if int(z["has_camera"]):
            c = z["camera"]
            s.camera = {"pos": c[0:3], "up": c[3:6], "right": c[6:9], "look_at": c[9:12]}
        for name, dt in _ARRAYS:
            setattr(s, name, z[name].astype(dt, copy=False))
        if "texture_names" in z.files:
            for i, nm in enumerate(z["texture_names"].tolist()):
                s.textures.append({"name": str(nm), "rgba": z[f"texture_{i}"] if f"texture_{i}" in z.files else None})
        return s

    def counts(self):
        return {name: int(getattr(self, name).shape[0]) for name, _ in _ARRAYS}

    @property
    def n_bvh_prims(self):
        return len(self.spheres) + len(self.triangles) + len(self.boxes) + len(self.smooth_triangles)

    def equals(self, other):
        if (self.camera is None) != (other.camera is None):
            return False
        if self.camera is not None:
            for k in ("pos", "up", "right", "look_at"):
                if np.asarray(self.camera[k], np.float32).tobytes() != np.asarray(other.camera[k], np.float32).tobytes():
                    return False
        if [t["name"] for t in self.textures] != [t["name"] for t in other.textures]:
            return False
        return all(getattr(self, n).tobytes() == getattr(other, n).tobytes() for n, _ in _ARRAYS)
