#!/usr/bin/env python3
"""Benchmark of the render path: Mrays/s (primary + secondary) and ms/frame of the bunny frame at 1920x1080.

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong]

Headline workload: the frame the metric names -- bunny_tasty.pov at 1920x1080, Cook-Torrance, 4x stratified
anti-aliasing (2x2 strata per pixel, hash jitter).  That scene file is not available, so the frame is rendered
on the reference's own bunny.pov (10 000 flat triangles, one light), the reference tree, every ray traced.

A step is one frame: one persistent traversal launch (primaries, bounces, shadow feelers) -> light sums ->
combine -> resolve, with the scene and BVH already resident in HBM and the 8-bit frame left in HBM.
`value` = rays of the frame / wall time per frame, `ms_per_step` = ms per frame.  The rays of the frame are the ones the
reference traces for it (kernel.cu:519-522 and the shade tree below); `config.primaries_answered_by_tile` says how many
of the primaries were answered as misses a whole tile at a time, because no ray through the tile can reach the scene's
bounding box (the reference answers each of them with its root's box test).  With N > 1 (one process per
GPU, launched by torch.distributed.run) the frame is sharded by 32x32-pixel tiles, tile k to rank k mod N,
the scene is replicated, and the compact tile buffers are gathered to rank 0 over RCCL and un-tiled there.
The gather of frame k overlaps the rendering of frame k + 1 (two part buffers); all K gathers and un-tiles
complete inside the timed region.

  --scaling weak (default)  the frame grows with N so that every GPU keeps a 1080p frame's worth of pixels of
                            the same view: 1920x1080, 3840x1080, 3840x2160, 7680x2160 for N = 1, 2, 4, 8
  --scaling strong          the 1920x1080 frame is split N ways

Objects that ride on the JSON line (rank 0; all but `roofline` and `valu` at N = 1 only):
  roofline      the dominant kernel (the persistent traversal kernel): algorithmic bytes per launch / its average
                launch duration (HIP events on the stream it runs on), against the ceiling of the memory level
                that serves those bytes: the XCDs' L2 while scene + BVH fit it, HBM otherwise.  Algorithmic bytes =
                what the reference's own walk of its own tree moves for these rays (per-ray figure from the CPU
                oracle's counters for this frame) x the rays the launch answers; what the kernel really fetched is
                reported beside it (`bytes_fetched_per_launch`)
  valu          the same kernel against the vector ALU: wave-instructions per launch (from the committed
                counter pass of this command) / (SIMDs x issue rate), and the algorithmic flop-equivalents
  cpu_baseline  the CPU oracle (a port of the reference's per-pixel logic) on this box's host cores, same frame
  secondary     balls2.pov 1920x1080 Phong (shadows + reflections) and the 1M-triangle / 256-light frame at
                3840x2160, 16 spp (opt-in surface-area tree; the reference's tree needs ~10^4 box tests per ray there)
  end_to_end    wall time of one cold launch_kernel-style call (host BVH build, upload, first frame, copy
                back, free) at 800x600, beside the reference's published whole-program times
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

WIDTH, HEIGHT, SAMPLES = 1920, 1080, 4
SCENE = "bunny"
WORKLOAD = ("bunny_tasty.pov 1920x1080 Cook-Torrance, 4x stratified AA (2x2 strata, hash jitter); the scene file is absent, "
            "rendered on stand-in A: the reference's bunny.pov (10 000 triangles, 1 light), reference BVH, depth 6")
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, GB/s
L2_PEAK_GBS = 34500.0          # aggregate L2 bandwidth of the 8 XCDs, GB/s
L2_BYTES_PER_XCD = 4 << 20
SIMDS = 256 * 4
VALU_ISSUE_PER_SIMD = 1.2e9    # one wave64 f32 instruction per 2 cycles at 2.4 GHz
FP32_PEAK_TFLOPS = 78.6        # non-packed f32 FMA peak (half the 157.3 TFLOP/s packed figure)
WEAK_FRAMES = {1: (1920, 1080), 2: (3840, 1080), 4: (3840, 2160), 8: (7680, 2160)}
DOMINANT = "k_trace<2, false"        # the uncounted whole-forest traversal kernel, as rocprofv3 names it (prefix)
PROFILE_TAG = "r02"
README_WALL_S = {"dragon_large_tris": 1.230, "balls2": 1.875}    # the reference's README:30-40, whole program, unnamed sm_20 GPU


def frame_size(world, scaling):
    if scaling == "strong" or world == 1:
        return WIDTH, HEIGHT
    if world in WEAK_FRAMES:
        return WEAK_FRAMES[world]
    return WIDTH * world, HEIGHT         # any other N: N frames' worth of columns over the same view


# What the reference's algorithm does for the headline frame (its own tree, its own walk: kernel.cu:83-200), from the
# CPU oracle's counters for exactly this frame (bunny 1920x1080, Cook-Torrance, 4 spp, hash jitter, seed 0); the
# cpu_baseline leg renders the same frame with the oracle and checks them.  Per ray: 33.1 first visits of a node, 10.4
# revisits after a pop, 0.92 primitive tests, 76.6 box tests.
REFERENCE_WORK = {"rays": 9315040, "node_first_visits": 307993442, "node_revisits": 97205010, "prim_tests": 8523320,
                  "box_tests": 713191894}


def reference_bytes_per_ray(work=REFERENCE_WORK):
    """Algorithmic bytes per ray: 64 per node record the reference's walk visits (a revisit after a pop counts again),
    64 per primitive record it tests, 32 for the ray read and 16 for the result written."""
    return (64.0 * (work["node_first_visits"] + work["node_revisits"]) + 64.0 * work["prim_tests"]) / work["rays"] + 48.0


def reference_flops_per_ray(work=REFERENCE_WORK):
    """Flop-equivalents per ray of the same walk: 60 per box test (a division counts 6), 90 per primitive test, 40 per ray."""
    return (60.0 * work["box_tests"] + 90.0 * work["prim_tests"]) / work["rays"] + 40.0


def algorithmic_bytes(stats):
    """Bytes of the records this library's traversal launch really fetched for a frame: 64 per BVH node record
    visited, 64 per primitive record tested, 32 per ray and 16 per result.  Less than the reference's walk needs:
    most rays are answered from a second tree or not walked at all.  Returns (extend, shadow)."""
    n_ext = stats["rays_primary"] + stats["rays_reflect"] + stats["rays_refract"]
    extend = 64 * stats["node_visits"] + 64 * stats["prim_tests"] + (32 + 16) * n_ext
    shadow = 64 * stats["node_visits_shadow"] + 64 * stats["prim_tests_shadow"] + (32 + 16) * stats["rays_shadow"]
    return extend, shadow


def algorithmic_flops(stats, shading, n_lights):
    """Flop-equivalents of the same work: 60 per box test (two per node record; a division counts 6), 90 per
    primitive test, 40 per ray, and per light per shade node 80 (Phong) or 140 (Cook-Torrance)."""
    import raytracer_amd as ra
    visits = stats["node_visits"] + stats["node_visits_shadow"]
    tests = stats["prim_tests"] + stats["prim_tests_shadow"]
    per_light = 140 if shading == ra.COOK_TORRANCE else 80
    return 60 * 2 * visits + 90 * tests + 40 * stats["rays_total"] + per_light * n_lights * stats["shade_nodes"]


def cpu_baseline(sd, width, height, samples, shading, jitter):
    """The oracle on the host cores: one full frame of the same workload, OpenMP over rows."""
    import orclib
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, int(os.environ.get("OMP_NUM_THREADS", "16") or 16), 16))
    osc = orclib.OracleScene(sd)
    t0 = time.perf_counter()
    r = osc.render(width, height, samples, shading, jitter, 0, 0, cores, want_samples=False, want_primary=False)
    dt = time.perf_counter() - t0
    c = r["counters"]
    rays = c["rays_primary"] + c["rays_shadow"] + c["rays_reflect"] + c["rays_refract"]
    work = {"rays": rays, "node_first_visits": c["node_first_visits"], "node_revisits": c["node_revisits"],
            "prim_tests": c["prim_tests"], "box_tests": c["box_tests"]}
    same = (width, height, samples) == (WIDTH, HEIGHT, SAMPLES) and all(int(work[k]) == REFERENCE_WORK[k] for k in REFERENCE_WORK)
    return {"value": round(rays / dt / 1e6, 3), "reference_work_matches_roofline_constants": bool(same),
            "reference_work": {k: int(v) for k, v in work.items()}, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "ms_per_frame": round(dt * 1e3, 1),
            "sample": f"one whole {SCENE} {width}x{height} frame at {samples} spp: {rays} rays in {dt:.2f} s wall on {cores} threads "
                      f"({dt * cores:.0f} core-seconds), OpenMP dynamic over rows"}


def profile_counter(pas, counter):
    """Per-launch value of a counter of the dominant kernel from the committed counter passes of this command
    (over the passes' timed steps, where the summary tells those apart from the warm-up frames)."""
    path = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_pmc_summary.json")
    try:
        # (the kernel has variants; the one the timed steps ran is the one the pass launched most often)
        best = None
        for kernel, counters in json.load(open(path))[pas].items():
            if kernel.startswith(DOMINANT) and (best is None or counters[counter]["launches"] > best["launches"]):
                best = counters[counter]
        return best.get("per_launch_timed_steps", best["per_launch"]) if best else None
    except Exception:
        pass
    return None


def best_of(scene, params, out_ptr, stream, frames):
    best = None
    for _ in range(frames):
        st = scene.render_device(params, out_ptr, stream)
        if best is None or st["ms_total"] < best["ms_total"]:
            best = st
    return best


def secondary_frames(ra, abi, torch, stream):
    """Frames of the other single-GPU configurations, outside the timed region: steady state, best of a few."""
    from raytracer_amd import synth
    out = {}
    sd = ra.SceneData.load(os.path.join(ROOT, "tests", "golden", "scenes", "balls2.npz"))
    buf = torch.zeros(3840 * 2160 * 4, dtype=torch.uint8, device="cuda")
    with ra.Scene(sd, device=0) as sc:
        p = ra.make_params(1920, 1080, 1, ra.PHONG, abi.RT_JITTER_HASH)
        for _ in range(3):
            sc.render_device(p, buf.data_ptr(), stream)
        st = best_of(sc, p, buf.data_ptr(), stream, 8)
    out["balls2_1080p"] = {"workload": "balls2.pov 1920x1080 Phong, 1 spp, shadows + reflections, reference BVH",
                           "ms_per_frame": round(st["ms_total"], 3), "Mrays_per_s": round(st["rays_total"] / st["ms_total"] / 1e3, 1),
                           "rays_per_frame": st["rays_total"]}
    t0 = time.perf_counter()
    label = ("1 000 000 procedural triangles, 256 point lights, 3840x2160, 16 spp, Phong; surface-area tree "
             "(RT_SCENE_FAST_BVH, outside the parity claim: the reference's tree needs ~10^4 box tests per ray on this scene)")
    try:
        big = synth.terrain(1000, 500)
        with ra.Scene(big, device=0, fast_bvh=True) as sc:
            info = sc.info()
            p = ra.make_params(3840, 2160, 16, ra.PHONG, abi.RT_JITTER_HASH)
            first = sc.render_device(p, buf.data_ptr(), stream)
            st = sc.render_device(p, buf.data_ptr(), stream)
        out["terrain_1M_256_lights_4k_16spp"] = {
            "workload": label, "ms_per_frame": round(st["ms_total"], 1), "first_frame_ms": round(first["ms_total"], 1),
            "Mrays_per_s": round(st["rays_total"] / st["ms_total"] / 1e3, 1), "rays_per_frame": st["rays_total"],
            "launches_per_frame": st["launches_trace"], "host_tree_build_ms": round(info["ms_build_host"], 0),
            "wall_s_incl_scene_generation_and_build": round(time.perf_counter() - t0, 1)}
    except Exception as e:      # reported, not hidden: the headline above does not depend on this frame
        out["terrain_1M_256_lights_4k_16spp"] = {"workload": label, "error": str(e)}
    return out


def frames_in_flight(ra, torch, sd, params, rays_per_frame, in_flight=2, frames=60):
    """Throughput of a frame sequence with `in_flight` frames in flight (rt_sequence_*: a replica of the scene, a
    stream and a host thread per frame in flight), so that a frame's last waves and its resolve pass overlap the next
    frame's first waves.  Outside the timed region; every frame is a whole frame."""
    bufs = [torch.zeros(params.width * params.height * 4, dtype=torch.uint8, device="cuda") for _ in range(in_flight)]
    with ra.Sequence(sd, device=0, frames_in_flight=in_flight) as seq:
        def run(n):
            for k in range(n):
                if k >= in_flight:
                    seq.wait()
                seq.submit(params, bufs[k % in_flight].data_ptr())
            for _ in range(min(n, in_flight)):
                seq.wait()
        run(8 * in_flight)                      # per replica: first frame, second-tree try-out, start order
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(frames)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return {"frames_in_flight": in_flight, "frames": frames, "ms_per_frame": round(dt / frames * 1e3, 4),
            "Mrays_per_s": round(rays_per_frame * frames / dt / 1e6, 1),
            "note": "a sequence of whole headline frames with two in flight (rt_sequence_*: two replicas of the scene, a host thread "
                    "and a stream each): the end of one frame's traversal launch, where few waves are left, and its resolve pass "
                    "overlap the start of the next frame's; `value` above is measured with one frame in flight"}


def end_to_end(ra, abi, first_ms, steady_ms):
    """One cold call in the shape of launch_kernel (kernel.cu:689-761): scene construction with the host BVH build,
    upload, first frame, copy back, free -- wall clock, at the reference's default 800x600, 1 spp, Phong."""
    out = {"headline_first_frame_ms": round(first_ms, 3), "headline_steady_frame_ms": round(steady_ms, 3), "cold_call_800x600": {}}
    for name in ("balls2", "dragon_large_tris"):
        sd = ra.SceneData.load(os.path.join(ROOT, "tests", "golden", "scenes", name + ".npz"))
        frame = np.zeros(800 * 600 * 4, np.uint8)
        t0 = time.perf_counter()
        ra.launch_kernel(sd, ra.PHONG, 800, 600, frame, 1)
        dt = time.perf_counter() - t0
        out["cold_call_800x600"][name] = {"wall_ms": round(dt * 1e3, 2),
                                          "reference_whole_program_s": README_WALL_S[name],
                                          "note": "reference figure: `time ./raytrace` incl. parse + TGA on an unnamed sm_20 GPU (README:30-40); context, not a like-for-like baseline"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=6)      # a scene tries out its four ways of using the second tree in frames 2-5 of a size
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the other configurations' frames, the fast-tree frame and the cold-call timing "
                                                                "(keeps a profile of this command to the benchmark's own launches)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N ranks share cuda:0 and exchange over gloo (host-staged): checks the N > 1 code path on a 1-GPU box")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import raytracer_amd as ra
    from raytracer_amd import abi, tiles

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N > 1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    sd = ra.SceneData.load(os.path.join(ROOT, "tests", "golden", "scenes", SCENE + ".npz"))
    shading, jitter = ra.COOK_TORRANCE, abi.RT_JITTER_HASH
    scene = ra.Scene(sd, device=local_rank)

    width, height = frame_size(world, args.scaling)
    part_bytes = tiles.part_bytes(width, height, world)
    layout = abi.RT_OUT_FRAME if world == 1 else abi.RT_OUT_TILES
    params = ra.make_params(width, height, SAMPLES, shading, jitter, 0, 0, rank, world, layout)
    frame = torch.zeros(height * width * 4, dtype=torch.uint8, device="cuda") if rank == 0 else None
    pipe = tiles.GatherPipeline(part_bytes, world, rank, "cuda", dst=0, host_staged=args.rehearse_on_one_gpu) if world > 1 else None
    part = pipe.part[0] if pipe else None
    stream = torch.cuda.current_stream().cuda_stream
    lib = abi.load_library()
    state = {"k": 0}

    def finish_frame(b):
        """Frame in buffer b: wait for its parts (stream-side), un-tile on rank 0."""
        parts = pipe.collect(b)
        if parts is not None and rank == 0:
            rc = lib.rt_untile_device(parts.data_ptr(), part_bytes, world, width, height, frame.data_ptr(), stream)
            if rc != 0:
                raise RuntimeError(lib.rt_last_error())

    def step():
        if world == 1:
            return scene.render_device(params, frame.data_ptr(), stream)
        # the one exchange, pipelined: frame k's compact tiles travel to rank 0 while frame k + 1 is rendered
        b = state["k"] % 2
        state["k"] += 1
        finish_frame(b)                           # the frame that used this buffer two steps ago
        st = scene.render_device(params, pipe.part[b].data_ptr(), stream)
        pipe.submit(b)
        return st

    def drain():
        if pipe:
            for b in (state["k"] % 2, (state["k"] + 1) % 2):
                finish_frame(b)

    def fence():
        drain()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    first_ms = None
    for _ in range(args.warmup):
        tw = time.perf_counter()
        step()
        if first_ms is None:
            torch.cuda.synchronize()
            first_ms = (time.perf_counter() - tw) * 1e3      # the scene's first frame: no tile order yet, queues still to be sized
    fence()
    t0 = time.perf_counter()
    acc = {}
    for _ in range(args.steps):
        st = step()
        for k in ("ms_total", "ms_trace"):
            acc[k] = acc.get(k, 0.0) + st[k]
    fence()
    dt = time.perf_counter() - t0

    red_dev = "cpu" if args.rehearse_on_one_gpu else "cuda"
    rays = torch.tensor([st["rays_total"], st["rays_primary"], st["rays_shadow"], st["rays_reflect"] + st["rays_refract"], st["primaries_culled"]],
                        dtype=torch.float64, device=red_dev)
    tmax = torch.tensor([dt], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    rays = rays.cpu().numpy()
    dt = float(tmax.item())
    ms_per_step = dt / args.steps * 1e3
    value = rays[0] * args.steps / dt / 1e6

    roofline = valu = baseline = secondary = e2e = None
    if rank == 0:
        # algorithmic bytes of the dominant kernel: one counted frame on the same tiles (outside the timed region)
        counted = ra.Scene(sd, device=local_rank, counters=True)
        info = counted.info()
        buf = part if world > 1 else frame
        cst = counted.render_device(params, buf.data_ptr(), stream)
        counted.close()
        b_extend, b_shadow = algorithmic_bytes(cst)
        # one persistent launch traces every ray of the frame (bounce rays and shadow feelers alike)
        launches = max(1, st["launches_trace"])
        avg_ms = acc["ms_trace"] / args.steps / launches
        # algorithmic bytes: the reference walk's per-ray figure x the rays this launch answers (the oracle counts the
        # feelers whose own-surface test missed as rays; so does this)
        rays_answered = cst["rays_total"] + cst["shadow_skipped"]
        bytes_per_launch = reference_bytes_per_ray() * rays_answered / launches
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        in_l2 = info["device_bytes"] <= L2_BYTES_PER_XCD
        peak = L2_PEAK_GBS if in_l2 else HBM_PEAK_GBS
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_trace_kernel.json")
        if os.path.exists(tpath) and world == 1:
            try:
                tj = json.load(open(tpath))
                traffic = tj.get("hbm_bytes_per_launch") if tj.get("tag") == PROFILE_TAG else None
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "level": "l2" if in_l2 else "hbm", "kernel": "k_trace<OUT_CHAIN>",
                    "achieved": round(achieved, 1), "peak": peak, "unit": "GB/s", "frac": round(achieved / peak, 4), "traffic": traffic,
                    "bytes_per_launch": int(bytes_per_launch), "avg_launch_ms": round(avg_ms, 4), "launches_per_step": launches,
                    "bytes_per_ray": round(reference_bytes_per_ray(), 1), "rays_per_launch": int(rays_answered / launches),
                    "bytes_fetched_per_launch": int((b_extend + b_shadow) / launches),
                    "bytes_fetched_closest_hit_rays": int(b_extend), "bytes_fetched_shadow_rays": int(b_shadow),
                    "scene_bytes": int(info["device_bytes"]), "hbm_peak": HBM_PEAK_GBS,
                    "note": ("memory-bound kernel; `achieved` = the reference walk's algorithmic bytes (64 per node record it visits and per "
                             "primitive it tests, 48 per ray: %.0f B/ray from the CPU oracle's counters for this frame) x rays answered / launch "
                             "time; `bytes_fetched_per_launch` is what this kernel's own counters say it fetched (it proves most answers "
                             "without the reference's walk); scene + trees (%.1f MB) fit each XCD's 4 MiB L2, so the records are served by L2 "
                             "and `peak` is the aggregate L2 bandwidth; `traffic` is the measured HBM-side bytes per launch"
                             % (reference_bytes_per_ray(), info["device_bytes"] / 1e6)) if in_l2 else
                            "scene + BVH exceed the XCDs' L2: `peak` is the HBM bandwidth"}
        insts = profile_counter("pmc_sq", "SQ_INSTS_VALU") if world == 1 else None
        flops_done = algorithmic_flops(cst, shading, len(sd.lights)) / launches
        flops = reference_flops_per_ray() * rays_answered / launches
        valu = {"kernel": "k_trace<OUT_CHAIN>", "wave_insts_per_launch": insts,
                "issue_peak_per_s": SIMDS * VALU_ISSUE_PER_SIMD,
                "frac": round(insts / (avg_ms * 1e-3) / (SIMDS * VALU_ISSUE_PER_SIMD), 4) if insts else None,
                "flop_equiv_per_launch": int(flops), "flop_equiv_done_per_launch": int(flops_done),
                "achieved_tflops_equiv": round(flops / (avg_ms * 1e-3) / 1e12, 2),
                "peak_tflops": FP32_PEAK_TFLOPS, "frac_flop_equiv": round(flops / (avg_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4),
                "lane_utilisation_node_steps": round((cst["node_visits"] + cst["node_visits_shadow"]) / max(1, 64 * cst["wave_node_steps"]), 3),
                "lane_utilisation_leaf_steps": round((cst["prim_tests"] + cst["prim_tests_shadow"]) / max(1, 64 * cst["wave_leaf_steps"]), 3),
                "note": "wave_insts_per_launch = SQ_INSTS_VALU of the committed counter pass of this command (profiles/); "
                        "issue peak = 1 024 SIMDs x one wave64 instruction per 2 cycles at 2.4 GHz; flop_equiv_per_launch = the "
                        "reference walk's flop-equivalents (60 per box test, 90 per primitive test, 40 per ray, from the oracle's "
                        "counters) x rays answered, flop_equiv_done_per_launch = the same formula on this kernel's own counts"}
        if world == 1 and not args.no_cpu_baseline:
            baseline = cpu_baseline(sd, width, height, SAMPLES, shading, jitter)
        if world == 1 and not args.no_secondary:
            secondary = secondary_frames(ra, abi, torch, stream)
            try:
                secondary["headline_two_frames_in_flight"] = frames_in_flight(ra, torch, sd, params, int(rays[0]))
            except Exception as e:      # reported, not hidden
                secondary["headline_two_frames_in_flight"] = {"error": str(e)}
            e2e = end_to_end(ra, abi, first_ms if first_ms is not None else 0.0, acc["ms_total"] / args.steps)

    if rank == 0:
        out = {
            "metric": "Mrays/s (primary+secondary); ms/frame bunny_tasty 1080p", "value": round(value, 2), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "ms_per_frame": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": WORKLOAD if (width, height) == (WIDTH, HEIGHT) else WORKLOAD.replace("1920x1080", f"{width}x{height}"),
                       "per_gpu": f"{width * height // world} pixels x {SAMPLES} samples",
                       "rays_per_frame": int(rays[0]), "primary": int(rays[1]), "shadow": int(rays[2]), "bounce": int(rays[3]),
                       "primaries_answered_by_tile": int(rays[4]),
                       "Mrays_per_s_without_those": round((rays[0] - rays[4]) * args.steps / dt / 1e6, 2),
                       "parallelism": f"tiles{world}", "tile": abi.RT_TILE, "jitter": "hash"},
            # HIP events of the timed steps: the frame on the caller's stream, k_trace on the stream it runs on; the kernels
            # behind it are timed one by one only on the instrumented scene (events between kernels cost stream time)
            "device_ms_per_step": dict({k: round(v / args.steps, 4) for k, v in acc.items()},
                                       ms_behind_trace=round((acc["ms_total"] - acc["ms_trace"]) / args.steps, 4),
                                       instrumented_frame={k: round(cst[k], 4) for k in ("ms_light", "ms_combine", "ms_resolve")})
            if rank == 0 else {k: round(v / args.steps, 4) for k, v in acc.items()},
            "roofline": roofline, "valu": valu, "cpu_baseline": baseline, "secondary": secondary, "end_to_end": e2e,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
