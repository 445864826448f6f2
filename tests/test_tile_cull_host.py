"""Which tiles a frame leaves out of its launch, checked on the host (no GPU).

rt_tiles_live says, per 32x32-pixel tile, whether any primary ray through it can reach a primitive's bounding box (first
the box that holds them all, then, down the tree, the primitives' own).  Here the rays are restated in float64 with
numpy from generateCameraRays (kernel.cu:477-498: origin pos + u/2 right + v/2 up, direction u/2 right + v/2 up +
normalize(lookAt - pos), with u, v in [-1, 1] over the frame and a jitter of at most half a sample spacing) and tested
against every primitive's box with the slab method: no ray of a tile that is left out may come anywhere near one, from
cameras near, far, inside, looking past the scene and looking away from it.
"""
import os

import numpy as np
import pytest

import raytracer_amd as ra
from raytracer_amd import abi, synth

T = abi.RT_TILE


def primitive_boxes(sd):
    """Bounds of every triangle of a scene made of untransformed triangles only: (n, 3) lows and highs."""
    p = [np.concatenate([np.asarray(sd.triangles[k], np.float64), np.asarray(sd.smooth_triangles[k], np.float64)]) for k in ("p1", "p2", "p3")]
    pts = np.stack(p, axis=1)                       # (n, 3 vertices, 3)
    return pts.min(1), pts.max(1)


def rays_of_tile(cam, width, height, spp_root, tx, ty, per_edge=9):
    """Origins and directions (float64) of a grid of rays over the tile's samples, jitter extremes included."""
    res_w, res_h = width * spp_root, height * spp_root
    ts = T * spp_root
    xs = np.linspace(tx * ts, min((tx + 1) * ts, res_w) - 1, per_edge)
    ys = np.linspace(ty * ts, min((ty + 1) * ts, res_h) - 1, per_edge)
    pos, up, right = (np.asarray(cam[k], np.float64) for k in ("pos", "up", "right"))
    look = np.asarray(cam["look_at"], np.float64) - pos
    look /= np.linalg.norm(look)
    o, d = [], []
    for jx in (-0.5, 0.5):
        for jy in (-0.5, 0.5):
            u = 2.0 * xs / res_w - 1.0 + jx / res_w
            v = 2.0 * ys / res_h - 1.0 + jy / res_h
            U, V = np.meshgrid(u, v, indexing="ij")
            off = 0.5 * U[..., None] * right + 0.5 * V[..., None] * up
            o.append((pos + off).reshape(-1, 3))
            d.append((off + look).reshape(-1, 3))
    return np.concatenate(o), np.concatenate(d)


def hits_box(o, d, lo, hi, grow):
    """Slab test of rays (t >= 0) against the box grown by `grow` on every side."""
    lo, hi = lo - grow, hi + grow
    with np.errstate(divide="ignore", invalid="ignore"):
        t1, t2 = (lo - o) / d, (hi - o) / d
    tmin, tmax = np.minimum(t1, t2), np.maximum(t1, t2)
    par = d == 0
    inside = (o >= lo) & (o <= hi)
    tmin = np.where(par, np.where(inside, -np.inf, np.inf), tmin)
    tmax = np.where(par, np.where(inside, np.inf, -np.inf), tmax)
    return (tmin.max(1) <= tmax.min(1)) & (tmax.min(1) >= 0.0)


VIEWS = [((0, 1, 9), (0, 0, 0)), ((0, 2, 30), (0, 0, 0)), ((14, 3, 9), (6, 0, 0)), ((0.1, 0.1, 0.2), (0, 0, -1)),
         ((0, 1, 9), (0, 1, 40)), ((0, 1, 40), (0, 1, 100)), ((-25, 10, 5), (0, 0, 0)), ((3, 30, 0.5), (0, 0, 0))]


@pytest.mark.parametrize("seed", range(4))
def test_no_ray_of_a_tile_left_out_comes_near_the_scene(seed):
    sd = synth.random_scene(40 + seed, n_spheres=0, n_boxes=0, n_planes=0, n_triangles=30, n_smooth=15)
    for arr in (sd.triangles, sd.smooth_triangles):
        arr["mod"]["trans"][:] = np.eye(4, dtype=np.float32).reshape(16)
        arr["mod"]["inv_trans"][:] = np.eye(4, dtype=np.float32).reshape(16)
    los, his = primitive_boxes(sd)
    some_left_out = False
    left_out, left_out_by_the_scene_box = 0, 0
    for pos, at in VIEWS:
        sd.camera = {"pos": np.array(pos, np.float32), "up": np.array([0, 1, 0], np.float32),
                     "right": np.array([1.3333, 0, 0], np.float32), "look_at": np.array(at, np.float32)}
        for (w, h, spp_root) in ((200, 150, 1), (333, 257, 2)):
            with ra.Scene(sd, host_only=True) as sc:
                live = sc.tiles_live(w, h, spp_root * spp_root)
            os.environ["RT_CULL_FINE"] = "0"
            try:
                with ra.Scene(sd, host_only=True) as sc:
                    coarse = sc.tiles_live(w, h, spp_root * spp_root)
            finally:
                os.environ.pop("RT_CULL_FINE", None)
            assert not (live & ~coarse).any()            # the finer test only ever leaves out more
            left_out += int((~live.astype(bool)).sum()); left_out_by_the_scene_box += int((~coarse.astype(bool)).sum())
            tiles_x, tiles_y = (w + T - 1) // T, (h + T - 1) // T
            assert len(live) == tiles_x * tiles_y
            for k, flag in enumerate(live):
                o, d = rays_of_tile(sd.camera, w, h, spp_root, k % tiles_x, k // tiles_x)
                if not flag:
                    some_left_out = True
                    for lo, hi in zip(los, his):
                        assert not hits_box(o, d, lo, hi, 2e-3).any(), (pos, at, w, h, k)
                else:
                    pass        # a live tile may well miss everything: the test errs on the side of tracing
    assert some_left_out
    assert left_out > left_out_by_the_scene_box      # sparse triangles: many tiles see the scene's box and no triangle's


def test_everything_is_live_with_planes_and_without_the_finer_test_from_inside(monkeypatch):
    sd = synth.random_scene(44, n_planes=0)
    sd.camera = {"pos": np.array([0.1, 0.1, 0.2], np.float32), "up": np.array([0, 1, 0], np.float32),
                 "right": np.array([1.3333, 0, 0], np.float32), "look_at": np.array([0, 0, -1], np.float32)}
    monkeypatch.setenv("RT_CULL_FINE", "0")
    with ra.Scene(sd, host_only=True) as sc:
        assert sc.tiles_live(320, 200).all()             # the camera sits inside the scene's box
    monkeypatch.delenv("RT_CULL_FINE")
    sd = synth.random_scene(44, n_planes=1)          # a plane is tested for every ray (kernel.cu:189-196)
    sd.camera["pos"] = np.array([0, 1, 40], np.float32)
    sd.camera["look_at"] = np.array([0, 1, 100], np.float32)
    with ra.Scene(sd, host_only=True) as sc:
        assert sc.tiles_live(320, 200).all()


def test_tile_parts_agree_with_the_whole_frame():
    sd = synth.random_scene(45, n_planes=0)
    sd.camera["pos"] = np.array([8, 2, 12], np.float32)
    with ra.Scene(sd, host_only=True) as sc:
        whole = sc.tiles_live(400, 300, 4)
        for first in range(3):
            part = sc.tiles_live(400, 300, 4, first, 3)
            assert np.array_equal(part, whole[first::3])
    assert 0 < whole.sum() < len(whole)
