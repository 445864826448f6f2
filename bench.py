#!/usr/bin/env python3
"""Benchmark of the render path: Mrays/s (primary + secondary) on balls2 at 1920x1080, Phong.

    python bench.py --gpus N --steps K --warmup W

A step is one frame: generate -> extend -> shadow -> shade (per depth level) -> combine -> resolve,
with the scene and BVH already resident in HBM and the 8-bit frame left in HBM.  With N > 1 (one
process per GPU, launched by torch.distributed.run) the frame is sharded by 32x32-pixel tiles, tile k
to rank k mod N, the scene is replicated, and the compact tile buffers are gathered to rank 0 over
RCCL and un-tiled there; that exchange is inside the timed region.  value = rays of the whole frame
/ max-over-ranks time.

Two extra objects ride on the JSON line:
  roofline      algorithmic bytes per launch of the dominant kernel (the shadow-ray traversal) divided
                by its average launch duration, measured with HIP events on the render stream
  cpu_baseline  the CPU oracle (a port of the reference's per-pixel logic) timed on this box's host
                cores on a bounded sample of the same workload (rank 0, N = 1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

WIDTH, HEIGHT, SAMPLES = 1920, 1080, 1
SCENE = "balls2"
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, GB/s


def algorithmic_bytes(stats, n_lights):
    """Bytes the traversal launches must move for the work they did (the flat layout's record sizes):
    64 per BVH node record fetched, 64 per primitive record tested + 48 for its inverse transform
    (balls2's spheres all carry one), 32 per ray read (the shade point is rebuilt from the 8-byte hit
    and the 32-byte ray) and 1 per shadow flag / 8 per hit written."""
    shadow = (64 * stats["node_visits_shadow"] + (64 + 48) * stats["prim_tests_shadow"]
              + (stats["rays_shadow"] + stats["shadow_skipped"]) * (8 + 32 + 64 + 48) + stats["shade_nodes"] * n_lights * 1)
    n_ext = stats["rays_primary"] + stats["rays_reflect"] + stats["rays_refract"]
    extend = (64 * stats["node_visits"] + (64 + 48) * stats["prim_tests"] + (n_ext - stats["rays_primary"]) * 32 + n_ext * 8)
    return shadow, extend


def cpu_baseline(sd, shading, jitter):
    """The oracle on the host cores, on a quarter-size frame of the same scene (same rays per pixel)."""
    import orclib
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, int(os.environ.get("OMP_NUM_THREADS", "16") or 16), 16))
    osc = orclib.OracleScene(sd)
    w, h = WIDTH // 2, HEIGHT // 2
    t0 = time.perf_counter()
    r = osc.render(w, h, SAMPLES, shading, jitter, 0, 0, cores, want_samples=False, want_primary=False)
    dt = time.perf_counter() - t0
    c = r["counters"]
    rays = c["rays_primary"] + c["rays_shadow"] + c["rays_reflect"] + c["rays_refract"]
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": f"{SCENE} {w}x{h} (a quarter of the frame's samples), {rays} rays in {dt:.2f} s, OpenMP over rows"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import raytracer_amd as ra
    from raytracer_amd import abi

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N > 1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    sd = ra.SceneData.load(os.path.join(ROOT, "tests", "golden", "scenes", SCENE + ".npz"))
    shading, jitter = ra.PHONG, abi.RT_JITTER_HASH
    scene = ra.Scene(sd, device=local_rank)
    n_lights = len(sd.lights)

    tiles_all = ra.tile_count(WIDTH, HEIGHT, 0, 1)
    tiles_mine = ra.tile_count(WIDTH, HEIGHT, rank, world)
    part_tiles = (tiles_all + world - 1) // world
    part_bytes = part_tiles * abi.RT_TILE * abi.RT_TILE * 4
    layout = abi.RT_OUT_FRAME if world == 1 else abi.RT_OUT_TILES
    params = ra.make_params(WIDTH, HEIGHT, SAMPLES, shading, jitter, 0, 0, rank, world, layout)
    frame = torch.zeros(HEIGHT * WIDTH * 4, dtype=torch.uint8, device="cuda")
    part = torch.zeros(part_bytes, dtype=torch.uint8, device="cuda") if world > 1 else None
    parts = torch.zeros(world * part_bytes, dtype=torch.uint8, device="cuda") if (world > 1 and rank == 0) else None
    gather_list = [parts[i * part_bytes:(i + 1) * part_bytes] for i in range(world)] if parts is not None else None
    stream = torch.cuda.current_stream().cuda_stream
    lib = abi.load_library()

    def step():
        if world == 1:
            return scene.render_device(params, frame.data_ptr(), stream)
        st = scene.render_device(params, part.data_ptr(), stream)
        dist.gather(part, gather_list, dst=0)
        if rank == 0:
            rc = lib.rt_untile_device(parts.data_ptr(), part_bytes, world, WIDTH, HEIGHT, frame.data_ptr(), stream)
            if rc != 0:
                raise RuntimeError(lib.rt_last_error())
        return st

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    acc = {}
    for _ in range(args.steps):
        st = step()
        for k in ("ms_total", "ms_extend", "ms_shadow", "ms_shade", "ms_combine", "ms_resolve"):
            acc[k] = acc.get(k, 0.0) + st[k]
    fence()
    dt = time.perf_counter() - t0

    rays = torch.tensor([st["rays_total"], st["rays_primary"], st["rays_shadow"], st["rays_reflect"] + st["rays_refract"]],
                        dtype=torch.float64, device="cuda")
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    rays = rays.cpu().numpy()
    dt = float(tmax.item())
    ms_per_step = dt / args.steps * 1e3
    value = rays[0] * args.steps / dt / 1e6

    roofline = None
    baseline = None
    if rank == 0:
        # algorithmic bytes of the dominant kernel: one counted frame on the same tiles (outside the timed region)
        counted = ra.Scene(sd, device=local_rank, counters=True)
        buf = part if world > 1 else frame
        cst = counted.render_device(params, buf.data_ptr(), stream)
        counted.close()
        b_shadow, b_extend = algorithmic_bytes(cst, n_lights)
        launches = max(1, st["launches_shadow"])
        avg_ms = acc["ms_shadow"] / args.steps / launches
        achieved = (b_shadow / launches) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_shadow_kernel.json")
        if os.path.exists(tpath) and world == 1:
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": "k_shadow", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "bytes_per_launch": int(b_shadow / launches), "avg_launch_ms": round(avg_ms, 4),
                    "launches_per_step": launches,
                    "extend_kernel": {"bytes_per_step": int(b_extend), "ms_per_step": round(acc["ms_extend"] / args.steps, 4)},
                    "note": "scene + BVH are ~1.3 MB and live in L2; achieved is algorithmic bytes over time, not HBM traffic"}
        if world == 1 and not args.no_cpu_baseline:
            baseline = cpu_baseline(sd, shading, jitter)

    if rank == 0:
        out = {
            "metric": "Mrays/s (primary+secondary)", "value": round(value, 2), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{SCENE}.pov {WIDTH}x{HEIGHT} Phong, {SAMPLES} spp, shadows + reflections, depth 6",
                       "rays_per_frame": int(rays[0]), "primary": int(rays[1]), "shadow": int(rays[2]), "bounce": int(rays[3]),
                       "parallelism": f"tiles{world}", "tile": abi.RT_TILE, "jitter": "hash"},
            "device_ms_per_step": {k: round(v / args.steps, 4) for k, v in acc.items()},
            "roofline": roofline, "cpu_baseline": baseline,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
