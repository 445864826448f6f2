"""CPU-side checks of the product: the C-ABI library, the host BVH builder, the parser and the TGA writer.

No compute entry point is called (there is no GPU here); what runs is host code of libraytracer_amd.so,
compared against the oracle and the committed fixtures.
"""
import ctypes as C
import os
import re

import numpy as np
import pytest

import orclib
import raytracer_amd as ra
from raytracer_amd import abi, tiles

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
ALL_SCENES = sorted(os.path.splitext(f)[0] for f in os.listdir(os.path.join(GOLD, "scenes")))


def scene(name):
    return ra.SceneData.load(os.path.join(GOLD, "scenes", name + ".npz"))


# ------------------------------------------------------------------ library
def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z_0-9]+|launch_kernel)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = abi.load_library()
    declared = set(declared_functions("raytracer_amd.h"))
    assert "launch_kernel" in declared_functions("raytracer_tokens.h")      # the C++ drop-in header declares only this one
    declared.add("launch_kernel")
    assert declared == set(abi.EXPORTED_SYMBOLS), declared ^ set(abi.EXPORTED_SYMBOLS)
    for name in declared:
        assert getattr(lib, name) is not None


def test_struct_sizes_match_header(tmp_path):
    """The ctypes mirrors against the C compiler's view of include/raytracer_amd.h: sizes of every struct that crosses
    the ABI, and the offset of every rt_stats field."""
    import subprocess
    fields = [n for n, _ in abi.Stats._fields_ if not n.startswith("pad")]
    prog = ['#include <stdio.h>', '#include <stddef.h>', '#include "raytracer_amd.h"', 'int main(void) {']
    for name in ("rt_camera", "rt_render_params", "rt_scene_info", "rt_stats", "rt_material", "rt_modifier", "rt_sphere", "rt_triangle",
                 "rt_box", "rt_smooth_triangle", "rt_plane", "rt_light", "rt_texture", "rt_scene_desc", "rt_bvh_node"):
        prog.append(f'printf("{name} %zu\\n", sizeof({name}));')
    for f in fields:
        prog.append(f'printf("rt_stats.{f} %zu\\n", offsetof(rt_stats, {f}));')
    prog += ['return 0; }']
    src = tmp_path / "sizes.c"
    src.write_text("\n".join(prog))
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    mirror = {"rt_camera": abi.Camera, "rt_render_params": abi.RenderParams, "rt_scene_info": abi.SceneInfo, "rt_stats": abi.Stats,
              "rt_texture": abi.Texture, "rt_scene_desc": abi.SceneDesc}
    for name, cls in mirror.items():
        assert C.sizeof(cls) == int(got[name]), name
    for name, dt in (("rt_material", abi.material_dtype), ("rt_modifier", abi.modifier_dtype), ("rt_sphere", abi.sphere_dtype),
                     ("rt_triangle", abi.triangle_dtype), ("rt_box", abi.box_dtype), ("rt_smooth_triangle", abi.smooth_triangle_dtype),
                     ("rt_plane", abi.plane_dtype), ("rt_light", abi.light_dtype), ("rt_bvh_node", abi.bvh_node_dtype)):
        assert dt.itemsize == int(got[name]), name
    for f in fields:
        assert getattr(abi.Stats, f).offset == int(got[f"rt_stats.{f}"]), f


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        abi.load_library(str(tmp_path / "libraytracer_amd.so"))


def test_render_without_device_is_an_error():
    if ra.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(ra.RaytracerError) as e:
        ra.Scene(scene("simple"))
    assert e.value.code == abi.RT_ERR_NO_DEVICE
    with ra.Scene(scene("simple"), host_only=True) as sc:
        with pytest.raises(ra.RaytracerError) as e:
            sc.render(32, 32)
        assert e.value.code == abi.RT_ERR_NO_DEVICE


def test_launch_kernel_mirror_rejects_bad_sample_count(capsys):
    out = np.full(16 * 16 * 4, 7, np.uint8)
    ra.launch_kernel(scene("simple"), ra.PHONG, 16, 16, out, 3)
    assert "Invalid sample count: 3" in capsys.readouterr().out        # kernel.cu:707-712
    assert (out == 7).all()


# ---------------------------------------------------------------------- BVH
@pytest.mark.parametrize("name", ALL_SCENES)
def test_host_bvh_equals_oracle_bvh(name):
    sd = scene(name)
    osc = orclib.OracleScene(sd)
    try:
        with ra.Scene(sd, host_only=True) as sc:
            info = sc.info()
            nodes, order = sc.bvh()
        onodes, oroot, odepth = osc.flatten()
        assert info["n_bvh_prims"] == osc.n_bvh_prims and info["n_planes"] == osc.n_planes
        assert info["root"] == oroot and info["tree_depth"] == odepth
        assert order.tolist() == osc.leaf_order().tolist()
        assert nodes.tobytes() == onodes.tobytes()
    finally:
        osc.close()


def test_bvh_shape_invariants():
    sd = scene("iceBot")
    with ra.Scene(sd, host_only=True) as sc:
        info = sc.info()
        nodes, order = sc.bvh()
    n = info["n_bvh_prims"]
    assert info["n_nodes"] == n - 1 and sorted(order.tolist()) == list(range(n))
    leaves = [~c for c in np.concatenate([nodes["left"], nodes["right"]]).tolist() if c < 0]
    assert sorted(leaves) == list(range(n))
    inner = [c for c in np.concatenate([nodes["left"], nodes["right"]]).tolist() if c >= 0]
    assert sorted(inner) == list(range(1, n - 1))          # pre-order numbering, root = 0, every node referenced once
    assert (nodes["lmin"] <= nodes["lmax"]).all() and (nodes["rmin"] <= nodes["rmax"]).all()


def test_degenerate_scenes_build():
    empty = ra.SceneData()
    empty.camera = scene("simple").camera
    with ra.Scene(empty, host_only=True) as sc:
        assert sc.info()["n_nodes"] == 0 and sc.info()["root"] == -2 ** 31
    one = scene("simple")
    assert one.n_bvh_prims == 1
    with ra.Scene(one, host_only=True) as sc:
        assert sc.info()["n_nodes"] == 0 and sc.info()["root"] == ~0


# ------------------------------------------------------------------- parser
@pytest.mark.parametrize("name", ["grammar_mix", "cr_only"])
def test_parser_on_committed_pov(name):
    got = ra.parse_pov(os.path.join(GOLD, "pov", name + ".pov"))
    want = ra.SceneData.load(os.path.join(GOLD, "pov", name + ".expected.npz"))
    assert got.equals(want)


@pytest.mark.parametrize("name", ["bad_object", "bad_vector"])
def test_parser_rejects_what_the_reference_rejects(name):
    with pytest.raises(ra.RaytracerError) as e:
        ra.parse_pov(os.path.join(GOLD, "pov", name + ".pov"))
    assert e.value.code == abi.RT_ERR_PARSE


def test_parser_missing_file():
    with pytest.raises(ra.RaytracerError) as e:
        ra.parse_pov(os.path.join(GOLD, "pov", "does_not_exist.pov"))
    assert e.value.code == abi.RT_ERR_IO


@pytest.mark.skipif(not os.path.isdir(orclib.REFERENCE_DIR), reason="reference checkout not present")
@pytest.mark.parametrize("name", [n for n in ALL_SCENES if n != "bunny_obj"])     # bunny_obj comes from an .obj, not a .pov
def test_parser_on_reference_scenes(name):
    """The scene fixtures were made by the reference's own parser; ours must produce the same bytes."""
    got = ra.parse_pov(os.path.join(orclib.REFERENCE_DIR, name + ".pov"))
    assert got.equals(scene(name))


def test_texture_names_and_coordinates_from_the_parser():
    got = ra.parse_pov(os.path.join(GOLD, "pov", "textured.pov"))
    assert [t["name"] for t in got.textures] == ["checker.bmp", "other.bmp"]
    assert got.triangles["mod"]["mat"]["tex_id"].tolist() == [0, 1, 0, -1]
    assert got.triangles["vt2"][0].tolist() == [1.0, 0.0] and got.triangles["vt3"][3].tolist() == [0.0, 0.0]
    assert got.smooth_triangles["vt3"][0].tolist() == [0.5, 1.0]
    if os.path.exists(os.path.join(GOLD, "pov", "textured.expected.npz")):
        assert got.equals(ra.SceneData.load(os.path.join(GOLD, "pov", "textured.expected.npz")))


def test_bmp_loader(tmp_path):
    """readBMP (kernel.cu:35-61): 54-byte header, bottom-up BGR rows -> r, g, b, 0 in file order."""
    w, h = 4, 2
    rng = np.random.default_rng(1)
    bgr = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    hdr = bytearray(54)
    hdr[0:2] = b"BM"
    hdr[10:14] = (54).to_bytes(4, "little"); hdr[14:18] = (40).to_bytes(4, "little")
    hdr[18:22] = w.to_bytes(4, "little"); hdr[22:26] = h.to_bytes(4, "little")
    hdr[26:28] = (1).to_bytes(2, "little"); hdr[28:30] = (24).to_bytes(2, "little")
    path = tmp_path / "t.bmp"
    path.write_bytes(bytes(hdr) + bgr.tobytes())
    px = ra.load_bmp(path)
    assert px.shape == (h, w, 4)
    assert (px[:, :, :3] == bgr[:, :, ::-1]).all() and (px[:, :, 3] == 0).all()
    with pytest.raises(ra.RaytracerError):
        ra.load_bmp(tmp_path / "missing.bmp")


def test_texture_reference_without_pixels_is_rejected():
    sd = scene("blitzcrank")
    sd.textures[0]["rgba"] = None
    with pytest.raises(ra.RaytracerError) as e:
        ra.Scene(sd, host_only=True)
    assert e.value.code == abi.RT_ERR_BAD_ARGUMENT


def test_scene_round_trip(tmp_path):
    sd = scene("cornell")
    p = str(tmp_path / "s.npz")
    sd.save(p)
    assert ra.SceneData.load(p).equals(sd)
    desc, keep = sd.to_desc()
    assert ra.SceneData.from_desc(desc).equals(sd)


# ---------------------------------------------------------------------- TGA
def test_tga_bytes(tmp_path):
    rng = np.random.default_rng(3)
    w, h = 37, 21
    frame = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    frame[0, :, 0] = np.arange(w) * 6 % 256
    frame.reshape(-1, 4)[:256, 1] = np.arange(256)           # every byte value goes through the channel map
    path = str(tmp_path / "a.tga")
    ra.write_tga(path, frame)
    b = open(path, "rb").read()
    assert len(b) == 18 + w * h * 3
    assert list(b[:18]) == [0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, w & 255, w >> 8, h & 255, h >> 8, 24, 0]
    lut = np.array([int(((k / 255.0) / 1.0) * 255) for k in range(256)], np.uint8)    # Image.cpp:77-101 in doubles
    want = lut[frame[:, :, 2::-1]]                                                       # BGR
    assert np.frombuffer(b, np.uint8, offset=18).reshape(h, w, 3).tobytes() == want.tobytes()
    if orclib.have_ref():
        path2 = str(tmp_path / "b.tga")
        orclib.ref().ref_write_tga(path2.encode(), w, h, frame.ctypes.data)
        assert open(path2, "rb").read() == b


def test_tga_bad_arguments(tmp_path):
    with pytest.raises(ra.RaytracerError):
        ra.write_tga(str(tmp_path / "no_such_dir" / "x.tga"), np.zeros((2, 2, 4), np.uint8))


# -------------------------------------------------------------------- tiles
@pytest.mark.parametrize("wh", [(1920, 1080), (640, 480), (33, 31), (1, 1), (32, 32), (100, 7)])
def test_tile_partition(wh):
    w, h = wh
    tx, ty = tiles.tile_grid(w, h)
    assert ra.tile_count(w, h, 0, 1) == tx * ty
    for world in (1, 2, 3, 8):
        seen = []
        for r in range(world):
            mine = tiles.tiles_of(w, h, r, world)
            assert ra.tile_count(w, h, r, world) == len(mine) <= tiles.part_tiles(w, h, world)
            seen += mine
        assert sorted(seen) == list(range(tx * ty))
    assert ra.tile_count(w, h, tx * ty, 1) == 0 and ra.tile_count(0, h, 0, 1) == 0


def test_untile_host_inverts_the_tile_layout():
    rng = np.random.default_rng(5)
    w, h, world = 100, 70, 3
    frame = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    stride = tiles.part_bytes(w, h, world)
    parts = np.zeros(world * stride, np.uint8)
    tx, _ = tiles.tile_grid(w, h)
    for r in range(world):
        for k, tile in enumerate(tiles.tiles_of(w, h, r, world)):
            x0, y0 = (tile % tx) * 32, (tile // tx) * 32
            blk = np.zeros((32, 32, 4), np.uint8)
            sub = frame[y0:y0 + 32, x0:x0 + 32]
            blk[:sub.shape[0], :sub.shape[1]] = sub
            parts[r * stride + k * 4096: r * stride + (k + 1) * 4096] = blk.reshape(-1)
    assert (tiles.untile_host(parts, stride, world, w, h) == frame).all()


@pytest.mark.skipif(not os.path.isdir(orclib.REFERENCE_DIR), reason="reference checkout not present")
def test_obj_import_reproduces_the_fixture():
    from raytracer_amd import synth
    assert synth.from_obj(os.path.join(orclib.REFERENCE_DIR, "bunny.obj")).equals(scene("bunny_obj"))
