"""Regenerates the committed fixtures under tests/golden/ (run in the build container only).

Inputs come from the reference checkout at /root/reference and from the reference's own parser /
classes compiled into oracle/_ref/libref.so (recipe: oracle/Makefile); nothing here reads or copies
reference source text.

  scenes/<name>.npz   the scene description the REFERENCE'S parser produces for <name>.pov
                      (flat arrays, see raytracer_amd/scene.py) -- input fixtures for parity tests
  tga/<name>.npz      pixels of the reference's own <name>.tga render (loose goldens: made by the
                      CUDA build with fast-math and cuRAND jitter)
  bvh_trace_2852.txt  the sort-range trace the reference keeps in its `output` file
  pov/*.expected.npz  what the reference's parser makes of the hand-written tests/golden/pov/*.pov
  classes.npz         known answers of the reference's primitive / shader classes on seeded inputs
"""
import ctypes as C
import glob
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import orclib
from raytracer_amd import synth  # noqa: E402
from raytracer_amd import abi  # noqa: E402

REF = orclib.REFERENCE_DIR
SCENES = ["simple", "simple_reflect", "simple_reflect2", "simple_refract", "simple_spec", "simple_tri", "simple_cam",
          "recurse_simp", "recurse_simp2", "transforms", "box", "final1", "final2", "cornell", "valentine",
          "ugly_part", "evil", "bunny_small_tris", "bunny", "dragon_large_tris", "balls2", "iceBot", "blitzcrank"]
# the rest of the reference's scene files (added later; `python make_fixtures.py --missing-scenes` writes only the scene
# fixtures that are not there yet and leaves every other fixture as it is)
SCENES += ["blitz", "donut", "gnarly", "miniShark", "monkey", "moo", "recurses", "recuse_simp2", "rgb", "rgb2", "rgbT", "shark",
           "sphere_bunny_small", "test", "test2", "test3"]
TGAS = ["recurse_simp2", "valentine", "final1", "simple_refract", "bunny", "dragon_large_tris", "balls2",
        # (the other images with a scene file of their name: recurse_simp and ugly_part agree with today's scene files -- 97 % of
        # the pixels within 1/255 --, the rest were made from earlier versions of their scenes: 0-52 %, not fixtures)
        "recurse_simp", "ugly_part"]
fp = orclib.fp


def read_tga(path):
    b = open(path, "rb").read()
    w, h = b[12] | b[13] << 8, b[14] | b[15] << 8
    bgr = np.frombuffer(b, np.uint8, w * h * 3, 18).reshape(h, w, 3)
    return np.ascontiguousarray(bgr[:, :, ::-1])      # rows bottom->top as stored, RGB


def read_bmp(path):
    """24-bit bottom-up .bmp -> (H, W, 4) uint8 (r, g, b, 0), rows in file order: what readBMP makes of it (kernel.cu:35-61)."""
    b = open(path, "rb").read()
    w, h = int.from_bytes(b[18:22], "little", signed=True), int.from_bytes(b[22:26], "little", signed=True)
    bgr = np.frombuffer(b, np.uint8, w * h * 3, 54).reshape(h, w, 3)
    out = np.zeros((h, w, 4), np.uint8)
    out[:, :, :3] = bgr[:, :, ::-1]
    return out


def scene_fixture(name):
    sd = orclib.ref_parse(os.path.join(REF, name + ".pov"))
    for t in sd.textures:
        # the reference binds blitz.bmp to every texture id, whatever the name (kernel.cu:580-598)
        t["rgba"] = read_bmp(os.path.join(REF, "blitz.bmp"))
    sd.save(os.path.join(HERE, "scenes", name + ".npz"))
    return sd.counts()


def missing_scenes():
    manifest = json.load(open(os.path.join(HERE, "MANIFEST.json")))
    for name in SCENES:
        if not os.path.exists(os.path.join(HERE, "scenes", name + ".npz")):
            manifest["scenes"][name] = scene_fixture(name)
            print(name, manifest["scenes"][name])
    for name in TGAS:
        if not os.path.exists(os.path.join(HERE, "tga", name + ".npz")):
            img = read_tga(os.path.join(REF, name + ".tga"))
            np.savez_compressed(os.path.join(HERE, "tga", name + ".npz"), rgb=img)
            manifest["tga"][name] = list(img.shape)
            print(name + ".tga", img.shape)
    json.dump(manifest, open(os.path.join(HERE, "MANIFEST.json"), "w"), indent=1, sort_keys=True)


def main():
    if "--missing-scenes" in sys.argv:
        return missing_scenes()
    manifest = {"scenes": {}, "tga": {}, "pov": {}}
    for name in SCENES:
        manifest["scenes"][name] = scene_fixture(name)
    # bunny.obj by the rules of the reference's objToPov.py, with a solid pigment (raytracer_amd.synth.from_obj)
    obj = synth.from_obj(os.path.join(REF, "bunny.obj"))
    obj.save(os.path.join(HERE, "scenes", "bunny_obj.npz"))
    manifest["scenes"]["bunny_obj"] = obj.counts()
    for name in TGAS:
        img = read_tga(os.path.join(REF, name + ".tga"))
        np.savez_compressed(os.path.join(HERE, "tga", name + ".npz"), rgb=img)
        manifest["tga"][name] = list(img.shape)
    with open(os.path.join(REF, "output")) as f:
        open(os.path.join(HERE, "bvh_trace_2852.txt"), "w").write(f.read())
    for pov in sorted(glob.glob(os.path.join(HERE, "pov", "*.pov"))):
        base = os.path.splitext(os.path.basename(pov))[0]
        try:
            sd = orclib.ref_parse(pov)
            sd.save(os.path.join(HERE, "pov", base + ".expected.npz"))
            manifest["pov"][base] = sd.counts()
        except RuntimeError as e:
            manifest["pov"][base] = {"error": str(e)}

    # known answers of the reference's classes on seeded inputs
    rng = np.random.default_rng(20261111)
    out = {}
    for name in ["transforms", "box", "cornell", "bunny_small_tris", "iceBot", "balls2", "blitzcrank"]:
        sd = orclib.ref_parse(os.path.join(REF, name + ".pov"))
        desc, keep = sd.to_desc()
        rs = orclib.ref().ref_scene_create(C.byref(desc))
        nprim = sd.n_bvh_prims + len(sd.planes)
        prims = rng.choice(nprim, min(nprim, 48), replace=False).astype(np.int32)
        c3 = np.zeros(3, np.float32)
        rays = np.zeros((len(prims), 8, 6), np.float32)
        ts = np.zeros((len(prims), 8), np.float32)
        nrm = np.zeros((len(prims), 8, 3), np.float32)
        uvs = np.zeros((len(prims), 8, 2), np.float32)
        bbox = np.zeros((len(prims), 6), np.float32)
        cen = np.zeros((len(prims), 3), np.float32)
        for i, p in enumerate(prims):
            p = int(p)
            orclib.ref().ref_prim_center(rs, p, c3.ctypes.data_as(fp))
            cen[i] = c3
            a, b = np.zeros(3, np.float32), np.zeros(3, np.float32)
            orclib.ref().ref_prim_bbox(rs, p, a.ctypes.data_as(fp), b.ctypes.data_as(fp))
            bbox[i, :3], bbox[i, 3:] = a, b
            for k in range(8):
                o = rng.normal(0, 5, 3).astype(np.float32)
                target = c3 if p < sd.n_bvh_prims else rng.normal(0, 3, 3).astype(np.float32)
                d = (target + rng.normal(0, 0.02, 3).astype(np.float32) - o).astype(np.float32)
                rays[i, k, :3], rays[i, k, 3:] = o, d
                t = orclib.ref().ref_prim_intersect(rs, p, o.ctypes.data_as(fp), d.ctypes.data_as(fp))
                ts[i, k] = t
                n = np.zeros(3, np.float32)
                uv = np.zeros(2, np.float32)
                if t > 0:
                    orclib.ref().ref_prim_normal(rs, p, o.ctypes.data_as(fp), d.ctypes.data_as(fp), C.c_float(t), n.ctypes.data_as(fp))
                    if name == "blitzcrank" and p >= len(sd.spheres) and p < sd.n_bvh_prims:   # the other classes print and return 0
                        orclib.ref().ref_prim_uv(rs, p, o.ctypes.data_as(fp), d.ctypes.data_as(fp), C.c_float(t), uv.ctypes.data_as(fp))
                nrm[i, k] = n
                uvs[i, k] = uv
        orclib.ref().ref_scene_destroy(rs)
        out.update({f"{name}_prims": prims, f"{name}_rays": rays, f"{name}_t": ts, f"{name}_normal": nrm, f"{name}_uv": uvs,
                    f"{name}_bbox": bbox, f"{name}_center": cen})
    # shaders
    n = 512
    sh_in = np.zeros((n, 20), np.float32)
    sh_out = np.zeros((2, n, 3), np.float32)
    for i in range(n):
        clr, lc = rng.random(3), rng.random(3) * 1.5
        amb, dif, spec = rng.random(3)
        rough = rng.choice([0.05, 0.01, 0.1, 0.3, 1.0, 10.0])
        v = rng.normal(size=(3, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
        row = np.concatenate([clr, [amb, dif, spec, rough], v[0], v[1], lc, v[2], [i % 3 == 0]]).astype(np.float32)
        sh_in[i] = row
        for st in (0, 1):
            o3 = np.zeros(3, np.float32)
            r = row
            orclib.ref().ref_shade(st, r[0:3].copy().ctypes.data_as(fp), C.c_float(r[3]), C.c_float(r[4]), C.c_float(r[5]), C.c_float(r[6]),
                                   r[7:10].copy().ctypes.data_as(fp), r[10:13].copy().ctypes.data_as(fp), r[13:16].copy().ctypes.data_as(fp),
                                   r[16:19].copy().ctypes.data_as(fp), int(r[19]), o3.ctypes.data_as(fp))
            sh_out[st, i] = o3
    out["shade_in"], out["shade_out"] = sh_in, sh_out
    np.savez_compressed(os.path.join(HERE, "classes.npz"), **out)
    json.dump(manifest, open(os.path.join(HERE, "MANIFEST.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(manifest, indent=1)[:600])


if __name__ == "__main__":
    main()
