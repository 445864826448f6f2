"""Procedural scenes for the scale configurations that have no .pov file: built in memory as SceneData,
handed to the C ABI like any parsed scene.  Host-side data generation only (numpy); nothing here renders.
"""
import numpy as np

from . import abi
from .scene import SceneData, identity_modifier


def terrain(n_x=1000, n_z=500, lights_per_side=16, seed=12345, extent=10.0):
    """A height field of 2*n_x*n_z flat triangles under a lights_per_side^2 grid of point lights.

    Quads over x, z in [-extent, extent], height 0.6 sin(0.9 x) cos(0.7 z) + 0.05 rand, two triangles per
    quad, identity transforms, finish ambient .2 diffuse .6 specular .3 roughness .05, every 8th triangle
    reflection .3; lights at y = 12 over the same extent, colour 1.5 / n_lights each; camera <0,9,18>
    looking at the origin, right <1.7778,0,0>.  The defaults are the 1M-triangle / 256-light case.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    xs = np.linspace(-extent, extent, n_x + 1, dtype=np.float64)
    zs = np.linspace(-extent, extent, n_z + 1, dtype=np.float64)
    X, Z = np.meshgrid(xs, zs, indexing="ij")
    Y = 0.6 * np.sin(0.9 * X) * np.cos(0.7 * Z) + 0.05 * rng.random(X.shape)
    V = np.stack([X, Y, Z], axis=-1).astype(np.float32)               # (n_x+1, n_z+1, 3)
    a, b, c, d = V[:-1, :-1], V[1:, :-1], V[1:, 1:], V[:-1, 1:]        # quad corners
    n_quads = n_x * n_z
    tris = np.zeros(2 * n_quads, abi.triangle_dtype)
    tris["p1"][0::2], tris["p2"][0::2], tris["p3"][0::2] = a.reshape(-1, 3), c.reshape(-1, 3), b.reshape(-1, 3)
    tris["p1"][1::2], tris["p2"][1::2], tris["p3"][1::2] = a.reshape(-1, 3), d.reshape(-1, 3), c.reshape(-1, 3)
    mod = identity_modifier(2 * n_quads)
    mod["mat"]["clr"] = np.stack([0.35 + 0.3 * rng.random(2 * n_quads), 0.55 + 0.3 * rng.random(2 * n_quads),
                                  0.3 + 0.2 * rng.random(2 * n_quads)], axis=-1).astype(np.float32)
    mod["mat"]["amb"], mod["mat"]["dif"], mod["mat"]["spec"], mod["mat"]["rough"] = 0.2, 0.6, 0.3, 0.05
    mod["mat"]["refl"][::8] = 0.3
    tris["mod"] = mod
    sd = SceneData()
    sd.triangles = tris
    n_l = lights_per_side * lights_per_side
    lights = np.zeros(n_l, abi.light_dtype)
    g = np.linspace(-extent, extent, lights_per_side, dtype=np.float32) if lights_per_side > 1 else np.zeros(1, np.float32)
    LX, LZ = np.meshgrid(g, g, indexing="ij")
    lights["pos"] = np.stack([LX.reshape(-1), np.full(n_l, 12.0, np.float32), LZ.reshape(-1)], axis=-1)
    lights["clr"] = np.float32(1.5 / n_l)
    sd.lights = lights
    sd.camera = {"pos": np.array([0, 9, 18], np.float32), "up": np.array([0, 1, 0], np.float32),
                 "right": np.array([1.7778, 0, 0], np.float32), "look_at": np.array([0, 0, 0], np.float32)}
    return sd


def glassy(scene: SceneData, everything=False):
    """A copy of `scene` whose triangles (or, with `everything`, all primitives) reflect and refract
    (reflection .3, refraction 1, ior 1.33, pigment rgbf <1,1,1,.8>): every hit spawns two rays, down to the
    depth limit -- the bounce-queue stress."""
    out = SceneData()
    out.camera = scene.camera
    for name in ("spheres", "triangles", "boxes", "smooth_triangles", "planes", "lights"):
        setattr(out, name, getattr(scene, name).copy())
    for name in ("triangles", "smooth_triangles") + (("spheres", "boxes", "planes") if everything else ()):
        arr = getattr(out, name)
        if len(arr):
            m = arr["mod"]["mat"]
            m["clr"], m["alpha"], m["refl"], m["refr"], m["ior"] = 1.0, 0.8, 0.3, 1.0, 1.33
    return out


def from_obj(path, pigment=(0.85, 0.8, 0.7)):
    """The scene the reference's objToPov.py makes of a Wavefront .obj (objToPov.py:7-23, 86-114), with a solid
    pigment where that script writes an image_map: one smooth_triangle per face from its first three corners
    (`v`, `vn` and, when present, `vt` indices, 1-based; coordinates pass through the "%.6f" text of a .pov
    file), finish ambient .2 diffuse .6 specular .4 roughness .05, camera <1,6,12> up <0,1,0> right <1.33,0,0>
    looking at the origin, one white light at <30,10,30>."""
    v, vn, vt, faces = [], [], [], []
    six = lambda x: float("%.6f" % float(x))
    with open(path, "r") as f:
        for line in f:
            w = line.split()
            if not w:
                continue
            if w[0] == "v":
                v.append([six(x) for x in w[1:4]])
            elif w[0] == "vn":
                vn.append([six(x) for x in w[1:4]])
            elif w[0] == "vt":
                vt.append([six(x) for x in w[1:3]])
            elif w[0] == "f":
                corners = [c.split("/") for c in w[1:4]]
                faces.append([(int(c[0]) - 1, int(c[1]) - 1 if len(c) == 3 and c[1] else -1, int(c[-1]) - 1) for c in corners])
    v, vn = np.asarray(v, np.float32), np.asarray(vn, np.float32)
    vt = np.asarray(vt, np.float32).reshape(-1, 2)
    fa = np.asarray(faces, np.int64)                                   # (n, 3 corners, (v, vt, vn))
    tris = np.zeros(len(fa), abi.smooth_triangle_dtype)
    for k, (pn, nn, tn) in enumerate((("p1", "n1", "vt1"), ("p2", "n2", "vt2"), ("p3", "n3", "vt3"))):
        tris[pn], tris[nn] = v[fa[:, k, 0]], vn[fa[:, k, 2]]
        if len(vt) and (fa[:, :, 1] >= 0).all():
            tris[tn] = vt[fa[:, k, 1]]
    mod = identity_modifier(len(fa))
    mod["mat"]["clr"] = np.asarray(pigment, np.float32)
    mod["mat"]["amb"], mod["mat"]["dif"], mod["mat"]["spec"], mod["mat"]["rough"] = 0.2, 0.6, 0.4, 0.05
    tris["mod"] = mod
    sd = SceneData()
    sd.smooth_triangles = tris
    lights = np.zeros(1, abi.light_dtype)
    lights["pos"], lights["clr"] = (30.0, 10.0, 30.0), (1.0, 1.0, 1.0)
    sd.lights = lights
    sd.camera = {"pos": np.array([1, 6, 12], np.float32), "up": np.array([0, 1, 0], np.float32),
                 "right": np.array([1.33, 0, 0], np.float32), "look_at": np.array([0, 0, 0], np.float32)}
    return sd


def random_scene(seed, n_spheres=12, n_triangles=10, n_boxes=5, n_smooth=6, n_planes=1, n_lights=2):
    """A small scene of every primitive class with random placement, random scale / rotate / translate modifiers on
    half of the objects (stored, like the reference's mat4, column-major, with the matching inverse) and random
    finishes, a third of them reflecting, a fifth refracting.  For parity runs over many seeds."""
    rng = np.random.Generator(np.random.PCG64(seed))

    def modifiers(n):
        mod = identity_modifier(n)
        m = mod["mat"]
        m["clr"] = rng.random((n, 3)).astype(np.float32)
        m["amb"], m["dif"] = 0.1 + 0.2 * rng.random(n), 0.3 + 0.5 * rng.random(n)
        m["spec"], m["rough"] = 0.6 * rng.random(n), 0.02 + 0.3 * rng.random(n)
        refl, refr = rng.random(n) < 1 / 3, rng.random(n) < 1 / 5
        m["refl"] = np.where(refl, 0.2 + 0.6 * rng.random(n), 0.0)
        m["refr"] = np.where(refr, 1.0, 0.0)
        m["alpha"] = np.where(refr, 0.3 + 0.6 * rng.random(n), 0.0)
        m["ior"] = np.where(refr, 1.1 + 0.6 * rng.random(n), 1.0)
        for k in range(n):
            if rng.random() < 0.5:
                continue
            sc = np.diag(np.append(0.5 + rng.random(3), 1.0))
            a, b = rng.random(2) * 2 * np.pi
            ry = np.array([[np.cos(a), 0, np.sin(a), 0], [0, 1, 0, 0], [-np.sin(a), 0, np.cos(a), 0], [0, 0, 0, 1]])
            rx = np.array([[1, 0, 0, 0], [0, np.cos(b), -np.sin(b), 0], [0, np.sin(b), np.cos(b), 0], [0, 0, 0, 1]])
            tr = np.eye(4); tr[:3, 3] = rng.random(3) - 0.5
            mat = (tr @ ry @ rx @ sc).astype(np.float32)
            mod["trans"][k] = mat.T.reshape(16)
            mod["inv_trans"][k] = np.linalg.inv(mat.astype(np.float64)).astype(np.float32).T.reshape(16)
        return mod

    def points(n, spread=4.0):
        return ((rng.random((n, 3)) - 0.5) * spread).astype(np.float32)

    sd = SceneData()
    sp = np.zeros(n_spheres, abi.sphere_dtype)
    sp["c"], sp["r"], sp["mod"] = points(n_spheres), (0.2 + 0.6 * rng.random(n_spheres)).astype(np.float32), modifiers(n_spheres)
    sd.spheres = sp
    tr = np.zeros(n_triangles, abi.triangle_dtype)
    base = points(n_triangles)
    tr["p1"], tr["p2"], tr["p3"] = base, base + points(n_triangles, 1.5), base + points(n_triangles, 1.5)
    tr["mod"] = modifiers(n_triangles)
    sd.triangles = tr
    bx = np.zeros(n_boxes, abi.box_dtype)
    lo = points(n_boxes)
    bx["p1"], bx["p2"], bx["mod"] = lo, lo + (0.2 + rng.random((n_boxes, 3))).astype(np.float32), modifiers(n_boxes)
    sd.boxes = bx
    st = np.zeros(n_smooth, abi.smooth_triangle_dtype)
    base = points(n_smooth)
    st["p1"], st["p2"], st["p3"] = base, base + points(n_smooth, 1.5), base + points(n_smooth, 1.5)
    for nn in ("n1", "n2", "n3"):
        v = rng.normal(size=(n_smooth, 3)); st[nn] = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
    st["mod"] = modifiers(n_smooth)
    sd.smooth_triangles = st
    pl = np.zeros(n_planes, abi.plane_dtype)
    pl["n"], pl["d"], pl["mod"] = (0.0, 1.0, 0.0), -2.5, modifiers(n_planes)
    sd.planes = pl
    lt = np.zeros(n_lights, abi.light_dtype)
    lt["pos"] = (points(n_lights, 6.0) + np.array([0, 6, 4], np.float32))
    lt["clr"] = (0.5 + 0.5 * rng.random((n_lights, 3))).astype(np.float32)
    sd.lights = lt
    sd.camera = {"pos": np.array([0, 1, 7], np.float32), "up": np.array([0, 1, 0], np.float32),
                 "right": np.array([1.3333, 0, 0], np.float32), "look_at": np.array([0, 0, 0], np.float32)}
    return sd


def similar(scene: SceneData, scale=1.0, shift=(0.0, 0.0, 0.0), move_camera=True, move_lights=True):
    """The scene under the similarity x -> scale * x + shift: composed onto every object's modifier (trans and inv_trans,
    worked out in double and rounded once each), and applied to the lights and the camera (its up / right vectors scale, so
    the picture is the same one) unless told to leave those where they are.  For parity runs far from the origin of the
    coordinates and at other sizes."""
    import copy
    out = copy.deepcopy(scene)
    shift = np.asarray(shift, np.float64)
    S = np.eye(4); S[:3, :3] *= scale; S[:3, 3] = shift
    Sinv = np.eye(4); Sinv[:3, :3] /= scale; Sinv[:3, 3] = -shift / scale
    for name in ("spheres", "triangles", "boxes", "smooth_triangles", "planes"):
        arr = getattr(out, name)
        for k in range(len(arr)):
            m = arr["mod"]["trans"][k].reshape(4, 4).T.astype(np.float64)          # stored column-major
            mi = arr["mod"]["inv_trans"][k].reshape(4, 4).T.astype(np.float64)
            arr["mod"]["trans"][k] = (S @ m).astype(np.float32).T.reshape(16)
            arr["mod"]["inv_trans"][k] = (mi @ Sinv).astype(np.float32).T.reshape(16)
    if move_lights and len(out.lights):
        out.lights["pos"] = (out.lights["pos"].astype(np.float64) * scale + shift).astype(np.float32)
    if move_camera and out.camera is not None:
        cam = out.camera
        out.camera = {"pos": (np.asarray(cam["pos"], np.float64) * scale + shift).astype(np.float32),
                      "look_at": (np.asarray(cam["look_at"], np.float64) * scale + shift).astype(np.float32),
                      "up": (np.asarray(cam["up"], np.float64) * scale).astype(np.float32),
                      "right": (np.asarray(cam["right"], np.float64) * scale).astype(np.float32)}
    return out
