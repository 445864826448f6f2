"""The benchmarked workloads at their full sizes: oracle parity plus size-independent properties.

balls2, 1920x1080:

  * oracle parity of the whole frame (primary ids + hit distances bit for bit, pixels within 1/255)
  * the frame is a pure function of (scene, params): two renders are the same bytes
  * tile-partition invariance: N parts rendered separately (what N ranks do) un-tile to the same bytes
  * the ray bookkeeping adds up, and the counted run sees the same rays as the uncounted one
the headline frame (bunny, 1920x1080, Cook-Torrance, 4 samples per pixel) in its warm state against the oracle, and the
glassy dragon at 3840x2160 through the size-independent properties.
"""
import os

import numpy as np
import pytest

import orclib
import raytracer_amd as ra
from raytracer_amd import abi, tiles

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
W, H = 1920, 1080


@pytest.fixture(scope="module")
def balls2():
    return ra.SceneData.load(os.path.join(GOLDEN, "scenes", "balls2.npz"))


@pytest.fixture(scope="module")
def full(balls2):
    with ra.Scene(balls2) as sc:
        frame, st = sc.render(W, H, 1, abi.RT_PHONG, abi.RT_JITTER_HASH)
        prim, t = sc.trace_primary(W, H, 1, abi.RT_JITTER_HASH)
    return frame, st, prim, t


def test_full_frame_against_oracle(balls2, full):
    frame, st, prim, t = full
    osc = orclib.OracleScene(balls2)
    ref = osc.render(W, H, 1, abi.RT_PHONG, abi.RT_JITTER_HASH, want_samples=False)
    osc.close()
    assert np.array_equal(prim, ref["prim"])
    assert np.array_equal(t.view(np.uint32), ref["t"].view(np.uint32))
    c = ref["counters"]
    assert (st["rays_primary"], st["rays_reflect"], st["rays_refract"]) == (c["rays_primary"], c["rays_reflect"], c["rays_refract"])
    assert st["rays_shadow"] + st["shadow_skipped"] == c["rays_shadow"]
    diff = np.abs(frame[:, :, :3].astype(int) - ref["rgba"][:, :, :3].astype(int)).max(axis=2)
    assert (diff <= 1).mean() >= 0.999, f"{(diff > 1).sum()} pixels off by more than 1"


def test_same_bytes_twice(balls2, full):
    with ra.Scene(balls2) as sc:
        a, st_a = sc.render(W, H, 1, abi.RT_PHONG, abi.RT_JITTER_HASH)
        b, st_b = sc.render(W, H, 1, abi.RT_PHONG, abi.RT_JITTER_HASH)
    assert a.tobytes() == b.tobytes() == full[0].tobytes()
    assert st_a["rays_total"] == st_b["rays_total"] == full[1]["rays_total"]


@pytest.mark.parametrize("world", [2, 8])
def test_tile_partition_invariance(balls2, full, world):
    import torch
    lib = abi.load_library()
    stride = tiles.part_bytes(W, H, world)
    parts = torch.zeros(world * stride, dtype=torch.uint8, device="cuda")
    frame = torch.zeros(H * W * 4, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    rays = 0
    with ra.Scene(balls2) as sc:
        for r in range(world):
            p = ra.make_params(W, H, 1, abi.RT_PHONG, abi.RT_JITTER_HASH, 0, 0, r, world, abi.RT_OUT_TILES)
            st = sc.render_device(p, parts.data_ptr() + r * stride, stream)
            assert st["samples"] <= ra.tile_count(W, H, r, world) * 32 * 32
            rays += st["rays_total"]
    assert lib.rt_untile_device(parts.data_ptr(), stride, world, W, H, frame.data_ptr(), stream) == 0
    torch.cuda.synchronize()
    got = frame.cpu().numpy().reshape(H, W, 4)
    assert got.tobytes() == full[0].tobytes()
    assert rays == full[1]["rays_total"]
    assert tiles.untile_host(parts.cpu().numpy(), stride, world, W, H).tobytes() == full[0].tobytes()


def test_ray_bookkeeping_and_counters(balls2, full):
    _, st, _, _ = full
    assert st["rays_primary"] == W * H == st["samples"]
    assert st["rays_total"] == st["rays_primary"] + st["rays_shadow"] + st["rays_reflect"] + st["rays_refract"]
    assert st["shade_nodes"] <= st["rays_primary"] + st["rays_reflect"] + st["rays_refract"]
    assert st["rays_shadow"] + st["shadow_skipped"] == st["shade_nodes"] * len(balls2.lights)
    with ra.Scene(balls2, counters=True) as sc:
        frame, cst = sc.render(W, H, 1, abi.RT_PHONG, abi.RT_JITTER_HASH)
    assert frame.tobytes() == full[0].tobytes()
    for k in ("rays_primary", "rays_shadow", "rays_reflect", "rays_refract", "shade_nodes"):
        assert cst[k] == st[k]
    assert cst["node_visits"] > cst["rays_primary"] and cst["prim_tests"] > 0 and cst["node_visits_shadow"] > 0


def test_antialiased_vga_frame_against_oracle():
    """simple.pov-sized AA frame: 640x480 at 4 samples per pixel on a mesh scene."""
    sd = ra.SceneData.load(os.path.join(GOLDEN, "scenes", "bunny.npz"))
    with ra.Scene(sd) as sc:
        frame, st = sc.render(640, 480, 4, abi.RT_PHONG, abi.RT_JITTER_HASH, seed=11)
    osc = orclib.OracleScene(sd)
    ref = osc.render(640, 480, 4, abi.RT_PHONG, abi.RT_JITTER_HASH, 11, want_samples=False, want_primary=False)
    osc.close()
    diff = np.abs(frame[:, :, :3].astype(int) - ref["rgba"][:, :, :3].astype(int)).max(axis=2)
    assert (diff <= 1).mean() >= 0.999
    assert st["rays_primary"] == 640 * 480 * 4


def test_headline_frame_against_oracle():
    """The frame the benchmark times -- bunny, 1920x1080, Cook-Torrance, 2x2 strata, hash jitter, seed 0 -- in the state it is
    timed in: the sixth frame of a scene, when the try-out of the second tree has settled and the tiles start in the
    order the earlier frames' costs give.  The hits that frame's own launch stored for its 8 294 400 primaries (tiles left
    out of the launch read as misses) are the oracle's bit for bit, the ray bookkeeping is equal and the pixels are within
    1/255 on at least 99.9 %."""
    sd = ra.SceneData.load(os.path.join(GOLDEN, "scenes", "bunny.npz"))
    spp, shading, jitter = 4, abi.RT_COOK_TORRANCE, abi.RT_JITTER_HASH
    with ra.Scene(sd) as sc:
        first = None
        for k in range(6):
            frame, st = sc.render(W, H, spp, shading, jitter, seed=0)
            first = frame if first is None else first
            assert frame.tobytes() == first.tobytes(), f"frame {k} differs from the scene's first frame"
        prim, t = sc.frame_primary_hits(W, H, spp)
    # the shortcuts the timed frame uses were in use here: tiles left out, answers proved from the second tree
    assert st["primaries_culled"] > 0 and st["second_tree"] != 0 and st["attempts"] == 1 and st["launches_trace"] == 1
    osc = orclib.OracleScene(sd)
    ref = osc.render(W, H, spp, shading, jitter, 0, want_samples=False)
    osc.close()
    assert np.array_equal(prim, ref["prim"])
    assert np.array_equal(t.view(np.uint32), ref["t"].view(np.uint32))
    c = ref["counters"]
    assert st["rays_primary"] == c["rays_primary"] == W * H * spp
    assert (st["rays_reflect"], st["rays_refract"], st["shade_nodes"]) == (c["rays_reflect"], c["rays_refract"], c["shade_nodes"])
    assert st["rays_shadow"] + st["shadow_skipped"] == c["rays_shadow"] and st["shadow_skipped"] == c["shadow_selfmiss"]
    diff = np.abs(frame[:, :, :3].astype(int) - ref["rgba"][:, :, :3].astype(int)).max(axis=2)
    assert (diff <= 1).mean() >= 0.999, f"{(diff > 1).sum()} pixels off by more than 1"
    assert (frame[:, :, 3] == 255).all()


def test_glassy_dragon_4k_properties():
    """The divergence-stress configuration at its own size: dragon_large_tris with every triangle reflecting and refracting,
    3840x2160, depth 6 (50 M rays; the oracle needs minutes for it, its parity is checked at 320x180 in test_gpu_scale.py).
    Properties that do not depend on size: the same bytes twice, 2- and 8-part tile partitions un-tile to the same bytes
    with the same ray total, the counting variant of the kernel renders the same frame and counts the same rays."""
    import torch
    from raytracer_amd import synth
    sd = synth.glassy(ra.SceneData.load(os.path.join(GOLDEN, "scenes", "dragon_large_tris.npz")))
    w, h = 3840, 2160
    lib = abi.load_library()
    stream = torch.cuda.current_stream().cuda_stream
    with ra.Scene(sd) as sc:
        whole, st = sc.render(w, h, 1, abi.RT_PHONG, abi.RT_JITTER_HASH)
        again, st2 = sc.render(w, h, 1, abi.RT_PHONG, abi.RT_JITTER_HASH)
        assert again.tobytes() == whole.tobytes() and st2["rays_total"] == st["rays_total"]
        assert st["rays_primary"] == w * h and st["rays_refract"] > 0 and st["rays_reflect"] > 0
        assert st["rays_total"] == st["rays_primary"] + st["rays_shadow"] + st["rays_reflect"] + st["rays_refract"]
        assert st["rays_shadow"] + st["shadow_skipped"] == st["shade_nodes"] * len(sd.lights)
        for world in (2, 8):
            stride = tiles.part_bytes(w, h, world)
            parts = torch.zeros(world * stride, dtype=torch.uint8, device="cuda")
            frame = torch.zeros(h * w * 4, dtype=torch.uint8, device="cuda")
            rays = 0
            for r in range(world):
                p = ra.make_params(w, h, 1, abi.RT_PHONG, abi.RT_JITTER_HASH, 0, 0, r, world, abi.RT_OUT_TILES)
                rays += sc.render_device(p, parts.data_ptr() + r * stride, stream)["rays_total"]
            assert lib.rt_untile_device(parts.data_ptr(), stride, world, w, h, frame.data_ptr(), stream) == 0
            torch.cuda.synchronize()
            assert frame.cpu().numpy().tobytes() == whole.tobytes(), world
            assert rays == st["rays_total"]
    with ra.Scene(sd, counters=True) as sc:
        counted, cst = sc.render(w, h, 1, abi.RT_PHONG, abi.RT_JITTER_HASH)
    assert counted.tobytes() == whole.tobytes()
    for k in ("rays_primary", "rays_shadow", "rays_reflect", "rays_refract", "shade_nodes", "shadow_skipped"):
        assert cst[k] == st[k], k


def test_chunked_frame_equals_unchunked(monkeypatch):
    """A frame cut into many chunks (more than the arena has slices for, so slices are re-used) is the same bytes."""
    sd = ra.SceneData.load(os.path.join(GOLDEN, "scenes", "cornell.npz"))
    with ra.Scene(sd) as sc:
        whole, st0 = sc.render(400, 300, 4, abi.RT_PHONG, abi.RT_JITTER_HASH)
        monkeypatch.setenv("RT_CHUNK_TILES", "7")
        parts, st1 = sc.render(400, 300, 4, abi.RT_PHONG, abi.RT_JITTER_HASH)
        monkeypatch.delenv("RT_CHUNK_TILES")
    assert st1["launches_trace"] > 8 and st0["launches_trace"] == 1
    assert parts.tobytes() == whole.tobytes()
    assert st0["rays_total"] == st1["rays_total"]


# the reference's other scene files with more than a handful of primitives, each at the size the benchmark is quoted at
# and in the state the benchmark measures (warm: try-out settled, tile order and finer tile test in force)
@pytest.mark.parametrize("name", ["dragon_large_tris", "iceBot", "gnarly", "shark", "miniShark", "monkey", "blitz", "blitzcrank",
                                  "sphere_bunny_small", "bunny_small_tris", "recurses", "donut", "cornell"])
def test_other_scenes_full_size_against_oracle(name):
    sd = ra.SceneData.load(os.path.join(GOLDEN, "scenes", name + ".npz"))
    with ra.Scene(sd) as sc:
        for _ in range(6):
            frame, st = sc.render(W, H, 1, abi.RT_PHONG, abi.RT_JITTER_HASH)
        prim, t = sc.trace_primary(W, H, 1, abi.RT_JITTER_HASH)
    osc = orclib.OracleScene(sd)
    ref = osc.render(W, H, 1, abi.RT_PHONG, abi.RT_JITTER_HASH, want_samples=False)
    osc.close()
    assert np.array_equal(prim, ref["prim"])
    assert np.array_equal(t.view(np.uint32), ref["t"].view(np.uint32))
    c = ref["counters"]
    assert (st["rays_primary"], st["rays_reflect"], st["rays_refract"]) == (c["rays_primary"], c["rays_reflect"], c["rays_refract"])
    assert st["rays_shadow"] + st["shadow_skipped"] == c["rays_shadow"]
    diff = np.abs(frame[:, :, :3].astype(int) - ref["rgba"][:, :, :3].astype(int)).max(axis=2)
    assert (diff <= 1).mean() >= 0.999, f"{name}: {(diff > 1).sum()} pixels off by more than 1"


@pytest.mark.parametrize("name", ["balls2", "iceBot", "gnarly", "dragon_large_tris", "blitzcrank"])
def test_other_scenes_full_size_cook_torrance_four_samples(name):
    """The headline's settings (Cook-Torrance, 2x2 strata, hash jitter) on other scenes, warm."""
    sd = ra.SceneData.load(os.path.join(GOLDEN, "scenes", name + ".npz"))
    with ra.Scene(sd) as sc:
        for _ in range(6):
            frame, st = sc.render(W, H, 4, abi.RT_COOK_TORRANCE, abi.RT_JITTER_HASH, seed=2)
        prim, t = sc.frame_primary_hits(W, H, 4)
    osc = orclib.OracleScene(sd)
    ref = osc.render(W, H, 4, abi.RT_COOK_TORRANCE, abi.RT_JITTER_HASH, 2, want_samples=False)
    osc.close()
    assert np.array_equal(prim, ref["prim"])
    assert np.array_equal(t.view(np.uint32), ref["t"].view(np.uint32))
    c = ref["counters"]
    assert (st["rays_primary"], st["rays_reflect"], st["rays_refract"]) == (c["rays_primary"], c["rays_reflect"], c["rays_refract"])
    assert st["rays_shadow"] + st["shadow_skipped"] == c["rays_shadow"]
    diff = np.abs(frame[:, :, :3].astype(int) - ref["rgba"][:, :, :3].astype(int)).max(axis=2)
    assert (diff <= 1).mean() >= 0.999, f"{name}: {(diff > 1).sum()} pixels off by more than 1"
