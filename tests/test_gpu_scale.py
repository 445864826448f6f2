"""Parity on the shapes of the larger configurations, at sizes the oracle finishes in seconds:

  * many lights: a 10 000-triangle height field under 256 point lights (257 rays per shaded sample)
  * full bounce trees: every surface both reflects and refracts, so a primary ray grows a binary tree of
    depth 6 and the traversal kernel's per-wave work list sees its worst case
  * Cook-Torrance with 4 strata per pixel on a mesh

Same bars as tests/test_gpu_parity.py.
"""
import os

import numpy as np
import pytest

import orclib
import raytracer_amd as ra
from raytracer_amd import abi, synth

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def compare(sd, w, h, samples=1, shading=abi.RT_PHONG, jitter=abi.RT_JITTER_HASH, seed=0, rgb_tol=2e-5):
    osc = orclib.OracleScene(sd)
    ref = osc.render(w, h, samples, shading, jitter, seed)
    osc.close()
    with ra.Scene(sd) as sc:
        prim, t = sc.trace_primary(w, h, samples, jitter, seed)
        rgb, _ = sc.render_samples(w, h, samples, shading, jitter, seed)
        frame, st = sc.render(w, h, samples, shading, jitter, seed)
        frame2, _ = sc.render(w, h, samples, shading, jitter, seed)
    assert np.array_equal(prim, ref["prim"])
    assert np.array_equal(t.view(np.uint32), ref["t"].view(np.uint32))
    c = ref["counters"]
    assert (st["rays_reflect"], st["rays_refract"], st["shade_nodes"]) == (c["rays_reflect"], c["rays_refract"], c["shade_nodes"])
    assert st["rays_shadow"] + st["shadow_skipped"] == c["rays_shadow"] and st["shadow_skipped"] == c["shadow_selfmiss"]
    both_nan = np.isnan(rgb) & np.isnan(ref["rgb"])
    d = np.where(both_nan, 0.0, np.abs(rgb - ref["rgb"]))
    assert not np.isnan(d).any() and d.max() <= rgb_tol, d.max()
    diff = np.abs(frame[:, :, :3].astype(int) - ref["rgba"][:, :, :3].astype(int)).max(axis=2)
    assert (diff <= 1).mean() >= 0.999
    assert frame.tobytes() == frame2.tobytes()
    return st


def test_256_lights():
    st = compare(synth.terrain(100, 50), 160, 90)
    assert st["rays_shadow"] > 200 * st["shade_nodes"]


def test_seventy_thousand_triangles_on_the_reference_tree():
    """A scene large enough for the host's threaded paths (per-primitive records filled from several threads, 72 200
    distinct materials through the hash table, radix-sorted rounds of the reference's tree build) against the oracle, whose
    scene construction is its own: ids, distances, ray counts and colours as for the small scenes."""
    sd = synth.terrain(190, 190, lights_per_side=2)
    with ra.Scene(sd, host_only=True) as sc:
        info = sc.info()
    assert info["n_bvh_prims"] == 72_200 and info["n_materials"] > 60_000
    st = compare(sd, 96, 72)
    assert st["rays_shadow"] > 0


def test_full_binary_bounce_trees():
    sd = synth.glassy(ra.SceneData.load(os.path.join(GOLDEN, "scenes", "dragon_large_tris.npz")))
    st = compare(sd, 320, 180)
    assert st["rays_reflect"] > 0 and st["rays_refract"] > 0
    assert st["rays_reflect"] + st["rays_refract"] > st["rays_primary"]


def test_closed_glass_room_grows_whole_trees():
    """A closed room in which everything reflects and refracts: few rays escape, so a primary grows a tree of about ten
    the whole 63-node tree."""
    sd = synth.glassy(ra.SceneData.load(os.path.join(GOLDEN, "scenes", "cornell.npz")), everything=True)
    st = compare(sd, 192, 144)
    assert st["rays_reflect"] + st["rays_refract"] > 8 * st["rays_primary"]


def test_spawn_list_overflow_is_retried(monkeypatch):
    """Lists that start far too short (64 entries per wave): the frame that overflows them is rendered again with
    longer ones and comes out the same bytes as with the default lists; the lists keep their length afterwards."""
    sd = synth.glassy(ra.SceneData.load(os.path.join(GOLDEN, "scenes", "cornell.npz")), everything=True)
    with ra.Scene(sd) as sc:
        want, st0 = sc.render(192, 144, 1)
    monkeypatch.setenv("RT_LIST_CAP", "64")
    with ra.Scene(sd) as sc:
        got, st = sc.render(192, 144, 1)
        again, st2 = sc.render(192, 144, 1)
    assert st["attempts"] > 1 and st2["attempts"] == 1
    assert st["max_spawn_list"] > 64
    assert got.tobytes() == want.tobytes() and again.tobytes() == want.tobytes()
    for k in ("rays_primary", "rays_shadow", "rays_reflect", "rays_refract", "shade_nodes"):
        assert st[k] == st0[k], k


def test_glassy_height_field():
    compare(synth.glassy(synth.terrain(60, 30, lights_per_side=2)), 256, 144)


def test_cook_torrance_four_strata():
    sd = ra.SceneData.load(os.path.join(GOLDEN, "scenes", "bunny.npz"))
    compare(sd, 480, 270, samples=4, shading=abi.RT_COOK_TORRANCE, seed=5, rgb_tol=1e-3)


def test_smooth_mesh_cook_torrance_four_strata():
    """bunny.obj as 4 968 smooth triangles (the reference's objToPov.py rules; camera moved in so that the mesh fills
    the frame): interpolated normals, Cook-Torrance, 2 x 2 strata -- the shape of the reference's headline frame."""
    sd = ra.SceneData.load(os.path.join(GOLDEN, "scenes", "bunny_obj.npz"))
    sd.camera = dict(sd.camera, pos=np.array([0.1, 0.7, 1.5], np.float32), look_at=np.array([-0.15, 0.35, 0.0], np.float32))
    st = compare(sd, 128, 96, samples=4, shading=abi.RT_COOK_TORRANCE, rgb_tol=1e-3)
    assert st["shade_nodes"] > 0.2 * st["rays_primary"]


def test_scheduling_knobs_do_not_change_a_pixel():
    """Tile start order, list draining order, refill threshold and burst length only schedule the work: frames rendered
    in child processes with each of them changed are the bytes of the frame rendered here."""
    import hashlib
    import subprocess
    import sys
    sd_path = os.path.join(GOLDEN, "scenes", "recurse_simp2.npz")
    with ra.Scene(ra.SceneData.load(sd_path)) as sc:
        sc.render(256, 192, 4)
        want, _ = sc.render(256, 192, 4)                 # the second frame uses the first one's tile costs
    digest = hashlib.sha256(want.tobytes()).hexdigest()
    code = ("import hashlib, sys, raytracer_amd as ra\n"
            "sc = ra.Scene(ra.SceneData.load(sys.argv[1]))\n"
            "sc.render(256, 192, 4); f, _ = sc.render(256, 192, 4)\n"
            "print(hashlib.sha256(f.tobytes()).hexdigest())\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for knobs in ({"RT_TILE_ORDER": "0"}, {"RT_TAKE_OLDEST": "0"}, {"RT_REFILL_IDLE": "40", "RT_NODE_BURST": "3"},
                  {"RT_CHUNK_TILES": "7"}):
        env = dict(os.environ, PYTHONPATH=root, **knobs)
        out = subprocess.run([sys.executable, "-c", code, sd_path], env=env, capture_output=True, text=True, timeout=100)
        assert out.returncode == 0, out.stderr
        assert out.stdout.strip().splitlines()[-1] == digest, knobs


@pytest.mark.parametrize("seed", range(12))
def test_random_scenes(seed):
    """Every primitive class, random modifiers and finishes (reflecting, refracting), both shaders by turns."""
    shading = abi.RT_PHONG if seed % 2 == 0 else abi.RT_COOK_TORRANCE
    compare(synth.random_scene(seed), 96, 72, samples=1 if seed % 3 else 4, shading=shading, seed=seed, rgb_tol=1e-3)


@pytest.mark.parametrize("seed", (100, 101, 102))
def test_random_scenes_with_deeper_trees(seed):
    sd = synth.random_scene(seed, n_spheres=300, n_triangles=200, n_boxes=60, n_smooth=100, n_planes=2, n_lights=3)
    compare(sd, 128, 96, shading=abi.RT_PHONG, seed=seed, rgb_tol=1e-3)


@pytest.mark.parametrize("seed", (7, 8))
def test_random_scenes_axis_aligned_view(seed):
    """Camera on the z axis, odd frame, no jitter: the centre column and row hold rays with a zero direction component,
    the case the box test treats as parallel to a slab (GeometryUtil.h:9-51)."""
    sd = synth.random_scene(seed, n_spheres=40, n_triangles=30, n_boxes=20, n_smooth=10)
    sd.camera = dict(sd.camera, pos=np.array([0, 0, 7], np.float32), look_at=np.array([0, 0, 0], np.float32))
    compare(sd, 65, 49, jitter=abi.RT_JITTER_NONE, rgb_tol=1e-3)


def test_million_triangles_256_lights_in_chunks():
    """The largest configuration's scene (1 000 000 triangles, 256 point lights) at a reduced frame, 4 samples per pixel,
    forced into several chunks of tiles.  The oracle cannot walk this scene in test time (its reference tree costs ~10^4
    box tests per ray here; the many-light comparison with the oracle is test_256_lights above, the nearest-hit check
    of this tree against brute force is in test_gpu_fast_bvh.py), so what is checked are the properties that do not
    depend on size: the same bytes twice, the same bytes whatever the chunking, the same bytes from 2 and 8 tile
    partitions, ray counts that add up, and the counting variant of the kernel producing the same frame."""
    sd = synth.terrain(1000, 500)
    w, h, spp = 256, 144, 4
    keep = {k: os.environ.get(k) for k in ("RT_CHUNK_TILES",)}
    try:
        with ra.Scene(sd, fast_bvh=True) as sc:
            info = sc.info()
            assert info["n_bvh_prims"] == 1_000_000 and info["n_lights"] == 256 and info["tree_depth"] > 14     # deeper than the LDS stack
            whole, st = sc.render(w, h, spp, abi.RT_PHONG, abi.RT_JITTER_HASH, seed=3)
            assert st["launches_trace"] == 1
            os.environ["RT_CHUNK_TILES"] = "7"
            chunked, st_c = sc.render(w, h, spp, abi.RT_PHONG, abi.RT_JITTER_HASH, seed=3)
            again, _ = sc.render(w, h, spp, abi.RT_PHONG, abi.RT_JITTER_HASH, seed=3)
            assert st_c["launches_trace"] == 6                     # 40 tiles in chunks of 7
            assert chunked.tobytes() == whole.tobytes() and again.tobytes() == whole.tobytes()
            for k in ("rays_primary", "rays_shadow", "rays_reflect", "shade_nodes", "shadow_skipped", "rays_total"):
                assert st_c[k] == st[k], k
            assert st["rays_shadow"] + st["shadow_skipped"] == 256 * st["shade_nodes"]
            assert st["rays_primary"] == w * h * spp and st["rays_reflect"] > 0
            for parts in (2, 8):
                total = np.zeros_like(whole)
                rays = 0
                for k in range(parts):
                    part, st_k = sc.render(w, h, spp, abi.RT_PHONG, abi.RT_JITTER_HASH, seed=3, tile_first=k, tile_stride=parts)
                    assert not (total[:, :, 3] & part[:, :, 3]).any()            # tiles are disjoint (alpha is 255 where written)
                    total |= part
                    rays += st_k["rays_total"]
                assert total.tobytes() == whole.tobytes()
                assert rays == st["rays_total"]
        os.environ.pop("RT_CHUNK_TILES", None)
        with ra.Scene(sd, fast_bvh=True, counters=True) as sc:
            counted, st_n = sc.render(w, h, spp, abi.RT_PHONG, abi.RT_JITTER_HASH, seed=3)
        assert counted.tobytes() == whole.tobytes()
        assert st_n["node_visits_shadow"] > 0 and st_n["rays_total"] == st["rays_total"]
        assert len(np.unique(whole[:, :, :3].reshape(-1, 3), axis=0)) > 1000
    finally:
        for k, v in keep.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_million_triangles_at_the_configuration_s_own_size():
    """The same scene at the size the configuration names: 3840x2160, 16 samples per pixel (16.7 G rays per frame, two
    launches).  Size-independent properties only (see above): the same bytes and counts whatever the chunking, counts that add up."""
    sd = synth.terrain(1000, 500)
    w, h, spp = 3840, 2160, 16
    keep = os.environ.get("RT_CHUNK_TILES")
    try:
        with ra.Scene(sd, fast_bvh=True) as sc:
            whole, st = sc.render(w, h, spp, abi.RT_PHONG, abi.RT_JITTER_HASH)
            os.environ["RT_CHUNK_TILES"] = "1500"
            chunked, st_c = sc.render(w, h, spp, abi.RT_PHONG, abi.RT_JITTER_HASH)
        assert st["launches_trace"] >= 2 and st_c["launches_trace"] == 6                     # 8 160 tiles in chunks of 1 500
        assert chunked.tobytes() == whole.tobytes()
        for k in ("rays_primary", "rays_shadow", "rays_reflect", "shade_nodes", "shadow_skipped", "rays_total"):
            assert st_c[k] == st[k], k
        assert st["rays_primary"] == w * h * spp
        assert st["rays_shadow"] + st["shadow_skipped"] == 256 * st["shade_nodes"]
        assert st["rays_total"] > 16_000_000_000
    finally:
        if keep is None:
            os.environ.pop("RT_CHUNK_TILES", None)
        else:
            os.environ["RT_CHUNK_TILES"] = keep
