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

  --scaling strong (default)  the 1920x1080 frame the metric names is split N ways (what the benchmark configuration
                              "bunny_tasty 1080p ... 1/2/4/8 GPUs image-tiled" asks for); with N > 1 a second timed region
                              then runs the weak frame and is reported under `weak`
  --scaling weak              the frame grows with N so that every GPU keeps a 1080p frame's worth of pixels of the same
                              view: 1920x1080, 3840x1080, 3840x2160, 7680x2160 for N = 1, 2, 4, 8 (strong under `strong`)

Objects that ride on the JSON line (rank 0; all but `roofline` and `valu` at N = 1 only):
  roofline      the dominant kernel (the persistent traversal kernel).  `achieved` = the bytes of the records THIS
                kernel's walk fetches per launch (64 per node record, 64 per primitive record, 48 per ray, from a
                counted render of the same frame) / its average launch duration (HIP events on the stream it runs
                on), against the ceiling of the memory level that serves them: the XCDs' L2 while scene + trees fit
                it, HBM otherwise.  `reference_equivalent` is the same formula on what the REFERENCE's walk of its
                own tree would fetch for these rays (the CPU oracle's counters): what the second tree, the
                certificates and the tiles left out save, not bandwidth the kernel reaches.  `served_by`, from the
                committed counter passes of this command: vector-L1 tag requests, the read requests that go on to
                L2, L2 hits and misses.  Wave-time shares and lane utilisation come from the counted frame, register
                spills from the compiler's resource remarks kept under profiles/
  valu          the same kernel against the vector ALU: wave-instructions per launch (from the committed
                counter pass of this command) / (SIMDs x issue rate), and the flop-equivalents done
  cpu_baseline  the CPU oracle (a port of the reference's per-pixel logic) on this box's host cores, same frame
  secondary     balls2.pov 1920x1080 Phong (shadows + reflections); the glassy dragon at 3840x2160, depth 6 (the
                stand-in for dragon_huge_tris.pov); the 1M-triangle / 256-light frame at 3840x2160, 16 spp (opt-in
                surface-area tree; the reference's tree needs ~10^4 box tests per ray there), with a roofline object
                of its own from a counted 1080p x 4 spp frame; the headline with two frames in flight
  predicted_strong_scaling   N = 1 only: every part of the headline frame cut 2, 4 and 8 ways rendered alone on this
                device, the slowest part + un-tile + the gather at one xGMI link per part: what tile sharding can
                reach, measured without the node
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
PROFILE_TAG = "r03"
GATHER64_PEAK_GBS = 14000.0    # what this part returns for divergent 64-byte fetches per lane from L2 (profiles/microbench/gather64.hip)
MALL_GATHER_GBS = 8600.0       # uniformly random rows from a table in the Infinity Cache
BIG_GATHER_GBS = 7650.0        # ... from a table of 150 MB and more (7.4-7.9 TB/s)
XGMI_LINK_GBS = 153.0          # one xGMI link, one direction
README_WALL_S = {"dragon_large_tris": 1.230, "balls2": 1.875}    # the reference's README:30-40, whole program, unnamed sm_20 GPU


DATA = ("the reference's own scene file bunny.pov (parsed by this library's parser into tests/golden/scenes/bunny.npz): stand-in A for "
        "bunny_tasty.pov, which is absent; nothing synthetic in the headline")


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


def cpu_baseline(sd, width, height, samples, shading, jitter, gpu_frame=None):
    """The oracle on the host cores: one full frame of the same workload, OpenMP over rows; its frame is compared with the
    frame the timed steps left on the device."""
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
    out = {"value": round(rays / dt / 1e6, 3), "reference_work_matches_roofline_constants": bool(same),
           "reference_work": {k: int(v) for k, v in work.items()}, "unit": "Mrays/s", "cores": cores, "kind": "port",
           "ms_per_frame": round(dt * 1e3, 1),
           "sample": f"one whole {SCENE} {width}x{height} frame at {samples} spp: {rays} rays in {dt:.2f} s wall on {cores} threads "
                     f"({dt * cores:.0f} core-seconds), OpenMP dynamic over rows"}
    if gpu_frame is not None:
        diff = np.abs(gpu_frame[:, :, :3].astype(np.int32) - r["rgba"][:, :, :3].astype(np.int32)).max(axis=2)
        out["frame_matches"] = round(float((diff <= 1).mean()), 6)
        out["frame_identical"] = round(float((diff == 0).mean()), 6)
        out["frame_matches_note"] = ("share of the pixels of the frame the timed steps left on the device that are within 1/255 per channel "
                                     "of this oracle frame (`frame_identical`: the same bytes)")
    return out


def profile_summary():
    try:
        return json.load(open(os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_pmc_summary.json")))
    except Exception:
        return {}


def profile_counter(pas, counter, prefix=DOMINANT, summary=None):
    """Per-launch value of a counter of the dominant kernel from the committed counter passes of this command
    (over the passes' timed steps, where the summary tells those apart from the warm-up frames)."""
    try:
        # (the kernel has variants; the one the timed steps ran is the one the pass launched most often)
        best = None
        for kernel, counters in (summary if summary is not None else profile_summary())[pas].items():
            if kernel.startswith(prefix) and counter in counters and (best is None or counters[counter]["launches"] > best["launches"]):
                best = counters[counter]
        return best.get("per_launch_timed_steps", best["per_launch"]) if best else None
    except Exception:
        return None


def served_by():
    """Where the traversal kernel's fetches are served, from the committed counter passes: requests at the vector L1's
    tags, the read requests the L1 passes on to L2, and L2's hits and misses (all per launch)."""
    sm = profile_summary()
    tag = [profile_counter("pmc_l1", f"TCP_TAGRAM{k}_REQ_sum", summary=sm) for k in range(4)]
    l1 = sum(t for t in tag if t is not None) if any(t is not None for t in tag) else None
    to_l2 = profile_counter("pmc_l2", "TCP_TCC_READ_REQ_sum", summary=sm)
    hit, miss = profile_counter("pmc_l2", "TCC_HIT_sum", summary=sm), profile_counter("pmc_l2", "TCC_MISS_sum", summary=sm)
    out = {"l1_tag_requests_per_launch": l1, "l1_read_requests_to_l2_per_launch": to_l2, "l2_hits_per_launch": hit, "l2_misses_per_launch": miss,
           "l2_requests_per_launch": profile_counter("pmc_l2", "TCC_REQ_sum", summary=sm)}
    if l1 and to_l2 is not None:
        out["share_served_by_vector_l1"] = round(1.0 - to_l2 / l1, 4)
    if hit is not None and miss is not None and hit + miss > 0:
        out["l2_hit_rate"] = round(hit / (hit + miss), 4)
    out["note"] = ("rocprofv3 --pmc passes of this command (profiles/%s_pmc_summary.json), timed steps; a tag request is one 128-byte line looked up "
                   "in the CU's vector L1, a request to L2 one 64-byte (or 128-byte) read it passes on" % PROFILE_TAG)
    return out


def kernel_spills():
    """Register spills of the timed instantiation of the traversal kernel, from the compiler's resource remarks kept under
    profiles/ (tools/kernel_resources.py)."""
    path = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_kernel_resources.txt")
    try:
        for line in open(path):
            if line.startswith("void k_trace<2, false, false, true>"):
                f = [x.strip() for x in line.split("|")]
                return {"kernel": f[0], "vgprs": int(f[1]), "vgpr_spills": int(f[3]), "sgpr_spills": int(f[4]), "scratch_bytes_per_lane": int(f[6]),
                        "waves_per_simd": int(f[7]), "source": os.path.relpath(path, ROOT)}
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


def counted_shape(ct):
    nv, npt = ct["node_visits"] + ct["node_visits_shadow"], ct["prim_tests"] + ct["prim_tests_shadow"]
    return {"record_fetches_per_ray": round((nv + npt) / max(1, ct["rays_total"]), 2),
            "lane_utilisation_node_steps": round(nv / max(1, 64 * ct["wave_node_steps"]), 3),
            "lane_utilisation_leaf_steps": round(npt / max(1, 64 * ct["wave_leaf_steps"]), 3),
            "wave_time_refill": round(ct["wave_time_refill"], 3), "wave_time_node": round(ct["wave_time_node"], 3), "wave_time_leaf": round(ct["wave_time_leaf"], 3),
            "hits_certified": ct["hits_certified"], "hits_contested": ct["hits_contested"], "hits_guided": ct["hits_guided"], "hits_unreached": ct["hits_unreached"],
            "shadows_certified": ct["shadows_certified"], "second_tree": ct["second_tree"],
            "wave_node_steps_with_rays_in_both_trees": round(ct["wave_steps_both_trees"] / max(1, ct["wave_node_steps"]), 3),
            "wave_node_steps_in_second_tree_only": round(ct["wave_steps_second_tree_only"] / max(1, ct["wave_node_steps"]), 3)}


def terrain_roofline(s4, c4, gbs, fetched, info):
    """The roofline object of the 1M-triangle scene's 1080p x 4 spp frame.  Which level serves the record fetches is read
    from the committed counter passes of that frame (profiles/<tag>_terrain_traffic.json): the ceiling is the one of the
    level that serves most of them."""
    path = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_terrain_traffic.json")
    t = json.load(open(path)) if os.path.exists(path) else {}
    hit_rate = t.get("l2_hit_rate")
    launch_s = s4["ms_trace"] / max(1, s4["launches_trace"]) * 1e-3
    from_l2 = t.get("l1_read_requests_to_l2_per_launch")
    in_l2 = hit_rate is not None and hit_rate >= 0.8
    peak = GATHER64_PEAK_GBS if in_l2 else MALL_GATHER_GBS
    return {"frame": "the same scene at 1920x1080, 4 spp (%.0f ms per frame, %d rays)" % (s4["ms_total"], s4["rays_total"]),
            "bound": "hbm", "level": "l2" if in_l2 else "infinity cache", "kernel": "k_trace<OUT_CHAIN> without the second tree",
            "achieved": round(gbs, 1), "peak": peak, "unit": "GB/s", "frac": round(gbs / peak, 4),
            "peak_from_the_infinity_cache": MALL_GATHER_GBS, "peak_beyond_the_cache": BIG_GATHER_GBS,
            "bytes_per_launch": int(fetched), "avg_launch_ms": round(launch_s * 1e3, 3),
            "ms_behind_trace": round(s4["ms_total"] - s4["ms_trace"], 3),
            "ms_light_sums_counted_frame": round(c4["ms_light"], 3), "ms_resolve_counted_frame": round(c4["ms_resolve"], 3),
            "traffic": t.get("hbm_bytes_per_launch"),
            "served_by": {"l1_tag_requests_per_launch": t.get("l1_tag_requests_per_launch"), "l1_read_requests_to_l2_per_launch": from_l2,
                          "l2_hit_rate": None if hit_rate is None else round(hit_rate, 4),
                          "GBs_out_of_l2": None if from_l2 is None else round(from_l2 * 64 / launch_s / 1e9, 1),
                          "GBs_fabric_side": None if t.get("hbm_bytes_per_launch") is None else round(t["hbm_bytes_per_launch"] / launch_s / 1e9, 1)},
            "note": "`achieved` = 64 bytes per node and primitive record this kernel's walk fetched + 48 per ray (counted frame) / the launch duration of the "
                    "uncounted one.  The scene's records (%.0f MB) do not fit the XCDs' L2, yet the counter passes of this frame (profiles/%s_terrain_*.json) "
                    "show where the fetches are served: neighbouring lanes walk the same nodes, so the CUs' vector L1 merges them (a tag request is one "
                    "128-byte line per instruction, however many lanes want it), L2 hits most of what is left, and `traffic` (fabric-side bytes per launch: "
                    "FETCH_SIZE doubled + WRITE_SIZE) is a few percent of `bytes_per_launch`.  `peak` is therefore this part's ceiling for divergent 64-byte "
                    "fetches out of L2 (profiles/microbench/gather64.hip); the guide's figures for random rows from the Infinity Cache and from beyond it are "
                    "beside it.  Like the headline, the launch is bound by its waves' dependent chains, not by bytes." % (info["device_bytes"] / 1e6, PROFILE_TAG)}


def secondary_frames(ra, abi, torch, stream):
    """Frames of the other single-GPU configurations, outside the timed region: steady state, best of a few."""
    from raytracer_amd import synth
    out = {}
    sd = ra.SceneData.load(os.path.join(ROOT, "tests", "golden", "scenes", "balls2.npz"))
    buf = torch.zeros(3840 * 2160 * 4, dtype=torch.uint8, device="cuda")
    with ra.Scene(sd, device=0) as sc:
        p = ra.make_params(1920, 1080, 1, ra.PHONG, abi.RT_JITTER_HASH)
        for _ in range(6):
            sc.render_device(p, buf.data_ptr(), stream)
        st = best_of(sc, p, buf.data_ptr(), stream, 8)
    out["balls2_1080p"] = {"workload": "balls2.pov 1920x1080 Phong, 1 spp, shadows + reflections, reference BVH",
                           "ms_per_frame": round(st["ms_total"], 3), "Mrays_per_s": round(st["rays_total"] / st["ms_total"] / 1e3, 1),
                           "rays_per_frame": st["rays_total"]}
    # the divergence stress: every triangle of dragon_large_tris.pov reflects and refracts (stand-in for dragon_huge_tris.pov)
    label = ("dragon_large_tris.pov with every triangle's finish rewritten to reflection 0.3 refraction 1.0 ior 1.33, pigment rgbf <1,1,1,0.8> "
             "(stand-in for the absent dragon_huge_tris.pov), 3840x2160, 1 spp, Phong, depth 6, reference BVH")
    try:
        gd = synth.glassy(ra.SceneData.load(os.path.join(ROOT, "tests", "golden", "scenes", "dragon_large_tris.npz")))
        p = ra.make_params(3840, 2160, 1, ra.PHONG, abi.RT_JITTER_HASH)
        with ra.Scene(gd, device=0) as sc:
            for _ in range(6):
                sc.render_device(p, buf.data_ptr(), stream)
            st = best_of(sc, p, buf.data_ptr(), stream, 4)
        with ra.Scene(gd, device=0, counters=True) as sc:
            for _ in range(6):
                ct = sc.render_device(p, buf.data_ptr(), stream)
        be, bs = algorithmic_bytes(ct)
        out["glassy_dragon_4k_depth6"] = dict({"workload": label, "ms_per_frame": round(st["ms_total"], 2), "ms_trace": round(st["ms_trace"], 2),
                                               "Mrays_per_s": round(st["rays_total"] / st["ms_total"] / 1e3, 1), "rays_per_frame": st["rays_total"],
                                               "bounce_rays": st["rays_reflect"] + st["rays_refract"], "attempts": st["attempts"],
                                               "bytes_fetched_per_frame": int(be + bs),
                                               "fetched_GBs": round((be + bs) / (st["ms_trace"] * 1e-3) / 1e9, 1)}, **counted_shape(ct))
    except Exception as e:      # reported, not hidden
        out["glassy_dragon_4k_depth6"] = {"workload": label, "error": str(e)}
    t0 = time.perf_counter()
    label = ("1 000 000 procedural triangles, 256 point lights, 3840x2160, 16 spp, Phong; surface-area tree "
             "(RT_SCENE_FAST_BVH, outside the parity claim: the reference's tree needs ~10^4 box tests per ray on this scene)")
    try:
        big = synth.terrain(1000, 500)
        t_gen = time.perf_counter() - t0
        with ra.Scene(big, device=0, fast_bvh=True) as sc:
            info = sc.info()
            p = ra.make_params(3840, 2160, 16, ra.PHONG, abi.RT_JITTER_HASH)
            first = sc.render_device(p, buf.data_ptr(), stream)
            st = sc.render_device(p, buf.data_ptr(), stream)
            # the same scene at 1080p x 4 spp (a frame that finishes in a quarter of a second), plain and counted: its roofline
            p4 = ra.make_params(1920, 1080, 4, ra.PHONG, abi.RT_JITTER_HASH)
            sc.render_device(p4, buf.data_ptr(), stream)
            s4 = best_of(sc, p4, buf.data_ptr(), stream, 2)
        with ra.Scene(big, device=0, fast_bvh=True, counters=True) as sc:
            c4 = sc.render_device(p4, buf.data_ptr(), stream)
        # the tree built on the device instead (Morton codes + radix tree): what the quicker build costs per frame
        with ra.Scene(big, device=0, gpu_bvh=True) as sc:
            info_g = sc.info()
            sc.render_device(p4, buf.data_ptr(), stream)
            g4 = best_of(sc, p4, buf.data_ptr(), stream, 2)
        be, bs = algorithmic_bytes(c4)
        fetched = (be + bs) / max(1, c4["launches_trace"])
        gbs = (be + bs) / (s4["ms_trace"] * 1e-3) / 1e9
        out["terrain_1M_256_lights_4k_16spp"] = {
            "workload": label, "ms_per_frame": round(st["ms_total"], 1), "first_frame_ms": round(first["ms_total"], 1),
            "Mrays_per_s": round(st["rays_total"] / st["ms_total"] / 1e3, 1), "rays_per_frame": st["rays_total"],
            "launches_per_frame": st["launches_trace"], "host_tree_build_ms": round(info["ms_build_host"], 0), "scene_generation_s": round(t_gen, 2),
            "scene_bytes": int(info["device_bytes"]),
            "device_built_tree": {"build_ms_host_side": round(info_g["ms_build_host"], 1), "build_ms_device_kernels": round(info_g["ms_build_device"], 2),
                                  "frame_ms_1080p_4spp": round(g4["ms_total"], 1), "frame_ms_1080p_4spp_surface_area_tree": round(s4["ms_total"], 1),
                                  "note": "RT_SCENE_GPU_BVH: the build drops from the host's %.0f ms to this; the frame costs the ratio of the two frame times "
                                          "more, so the host's surface-area tree pays for itself after a few frames of this size" % info["ms_build_host"]},
            "roofline": dict(terrain_roofline(s4, c4, gbs, fetched, info),
                             **counted_shape(c4)),
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


def predicted_strong_scaling(ra, abi, tiles, torch, sd, shading, jitter, stream, one_gpu_ms):
    """What sharding the headline frame by tiles over N devices can reach, measured on this one device: every part of the
    frame cut N ways rendered alone (a scene of its own, warmed like the headline: try-out of the second tree, start
    order), the slowest part's frame time, plus the un-tile pass on the gathering device, plus the gather of N - 1 parts
    over one xGMI link each (they arrive side by side).  Scene and trees are replicated, so nothing else is exchanged."""
    lib = abi.load_library()
    out = {"frame": f"{WIDTH}x{HEIGHT}, {SAMPLES} spp", "one_device_ms": round(one_gpu_ms, 4), "by_devices": {}}
    frame = torch.zeros(HEIGHT * WIDTH * 4, dtype=torch.uint8, device="cuda")
    xs, ys = [], []
    for n in (2, 4, 8):
        stride = tiles.part_bytes(WIDTH, HEIGHT, n)
        parts = torch.zeros(n * stride, dtype=torch.uint8, device="cuda")
        ms, live = [], []
        for k in range(n):
            p = ra.make_params(WIDTH, HEIGHT, SAMPLES, shading, jitter, 0, 0, k, n, abi.RT_OUT_TILES)
            with ra.Scene(sd, device=0) as sc:
                for _ in range(6):
                    sc.render_device(p, parts.data_ptr() + k * stride, stream)
                st = best_of(sc, p, parts.data_ptr() + k * stride, stream, 6)
                live.append(int(sc.tiles_live(WIDTH, HEIGHT, SAMPLES, k, n).sum()))
            ms.append(st["ms_total"])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        lib.rt_untile_device(parts.data_ptr(), stride, n, WIDTH, HEIGHT, frame.data_ptr(), stream)
        e0.record()
        for _ in range(10):
            lib.rt_untile_device(parts.data_ptr(), stride, n, WIDTH, HEIGHT, frame.data_ptr(), stream)
        e1.record()
        torch.cuda.synchronize()
        untile_ms = e0.elapsed_time(e1) / 10
        gather_ms = stride / (XGMI_LINK_GBS * 1e9) * 1e3
        frame_ms = max(ms) + untile_ms + gather_ms
        out["by_devices"][str(n)] = {"part_ms_slowest": round(max(ms), 4), "part_ms_mean": round(sum(ms) / n, 4), "slowest_over_mean": round(max(ms) / (sum(ms) / n), 3),
                                     "live_tiles_per_part": live, "untile_ms": round(untile_ms, 4), "gather_ms_at_one_link_per_part": round(gather_ms, 4),
                                     "predicted_frame_ms": round(frame_ms, 4), "predicted_speedup": round(one_gpu_ms / frame_ms, 3),
                                     "predicted_efficiency": round(one_gpu_ms / frame_ms / n, 3)}
        xs.append(1.0 / n); ys.append(sum(ms) / n)
    # part time against the share of the frame a part holds: the intercept is what a part costs however little it renders
    xs.append(1.0); ys.append(one_gpu_ms)
    a = np.polyfit(np.array(xs), np.array(ys), 1)
    out["fixed_cost_per_part_ms"] = round(float(a[1]), 4)
    out["note"] = ("measured on ONE device, no multi-GPU run behind it: part times are real frames of the parts' tiles; the gather is priced at %.0f GB/s per "
                   "link with the parts arriving side by side; fixed cost = the intercept of mean part time against 1/N (launch set-up, the "
                   "start of the persistent launch, the resolve pass, the host's wake-up)" % XGMI_LINK_GBS)
    return out


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
        # the same call taken apart (a second scene of the same file: the library and the device are warm by now, as they
        # are for the call above after the headline frames)
        t1 = time.perf_counter()
        sc = ra.Scene(sd)
        t2 = time.perf_counter()
        info = sc.info()
        _, st = sc.render(800, 600, 1, ra.PHONG, abi.RT_JITTER_HASH)
        t3 = time.perf_counter()
        sc.close()
        t4 = time.perf_counter()
        out["cold_call_800x600"][name] = {"wall_ms": round(dt * 1e3, 2),
                                          "parts_ms": {"scene_create": round((t2 - t1) * 1e3, 2), "of_it_records_and_first_tree": round(info["ms_build_host"], 2),
                                                       "of_it_upload_allocations_and_wait_for_second_tree": round(info["ms_upload"], 2),
                                                       "first_frame_call_with_copy_back": round((t3 - t2) * 1e3, 2), "of_it_on_the_device": round(st["ms_total"], 3),
                                                       "scene_destroy": round((t4 - t3) * 1e3, 2)},
                                          "reference_whole_program_s": README_WALL_S[name],
                                          "note": "reference figure: `time ./raytrace` incl. parse + TGA on an unnamed sm_20 GPU (README:30-40); context, not a like-for-like baseline"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=6)      # a scene tries out its four ways of using the second tree in frames 2-5 of a size
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the other configurations' frames, the fast-tree frame, the predicted scaling and the "
                                                                "cold-call timing (keeps a profile of this command to the benchmark's own launches)")
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
    stream = torch.cuda.current_stream().cuda_stream
    lib = abi.load_library()
    red_dev = "cpu" if args.rehearse_on_one_gpu else "cuda"

    def timed_region(scaling):
        """Warm-up, barrier, `steps` frames, barrier: returns what rank 0 reports for this frame size."""
        scene = ra.Scene(sd, device=local_rank)
        width, height = frame_size(world, scaling)
        part_bytes = tiles.part_bytes(width, height, world)
        layout = abi.RT_OUT_FRAME if world == 1 else abi.RT_OUT_TILES
        params = ra.make_params(width, height, SAMPLES, shading, jitter, 0, 0, rank, world, layout)
        frame = torch.zeros(height * width * 4, dtype=torch.uint8, device="cuda") if rank == 0 else None
        pipe = tiles.GatherPipeline(part_bytes, world, rank, "cuda", dst=0, host_staged=args.rehearse_on_one_gpu) if world > 1 else None
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

        def fence():
            if pipe:
                for b in (state["k"] % 2, (state["k"] + 1) % 2):
                    finish_frame(b)
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
        rays = torch.tensor([st["rays_total"], st["rays_primary"], st["rays_shadow"], st["rays_reflect"] + st["rays_refract"], st["primaries_culled"]],
                            dtype=torch.float64, device=red_dev)
        tmax = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        if world > 1:
            dist.all_reduce(rays, op=dist.ReduceOp.SUM)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        return {"scene": scene, "params": params, "frame": frame, "part": pipe.part[0] if pipe else None, "width": width, "height": height,
                "rays": rays.cpu().numpy(), "dt": float(tmax.item()), "acc": acc, "st": st, "first_ms": first_ms}

    R = timed_region(args.scaling)
    rays, dt, acc, st, width, height = R["rays"], R["dt"], R["acc"], R["st"], R["width"], R["height"]
    ms_per_step = dt / args.steps * 1e3
    value = rays[0] * args.steps / dt / 1e6
    other = None
    if world > 1:
        # the other way of scaling, in a timed region of its own
        which = "weak" if args.scaling == "strong" else "strong"
        R["scene"].close()
        O = timed_region(which)
        other = {"scaling": which, "frame": f"{O['width']}x{O['height']}", "value": round(O["rays"][0] * args.steps / O["dt"] / 1e6, 2), "unit": "Mrays/s",
                 "ms_per_step": round(O["dt"] / args.steps * 1e3, 4), "rays_per_frame": int(O["rays"][0])}
        O["scene"].close()

    roofline = valu = baseline = secondary = e2e = predicted = None
    if rank == 0:
        # the bytes the dominant kernel fetches: one counted frame on the same tiles (outside the timed region)
        params, frame, part = R["params"], R["frame"], R["part"]
        counted = ra.Scene(sd, device=local_rank, counters=True)
        info = counted.info()
        buf = part if world > 1 else frame
        gpu_frame = frame.cpu().numpy().reshape(height, width, 4).copy() if world == 1 else None
        for _ in range(6):          # (like the timed scene: past the try-out of the second tree, with a start order)
            cst = counted.render_device(params, buf.data_ptr(), stream)
        counted.close()
        b_extend, b_shadow = algorithmic_bytes(cst)
        # one persistent launch traces every ray of the frame (bounce rays and shadow feelers alike)
        launches = max(1, st["launches_trace"])
        avg_ms = acc["ms_trace"] / args.steps / launches
        fetched = (b_extend + b_shadow) / launches
        achieved = fetched / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # what the reference's walk would fetch for the rays this launch answers (the oracle counts the feelers whose
        # own-surface test missed as rays; so does this)
        rays_answered = cst["rays_total"] + cst["shadow_skipped"]
        ref_bytes = reference_bytes_per_ray() * rays_answered / launches
        ref_gbs = ref_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
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
                    "bytes_per_launch": int(fetched), "avg_launch_ms": round(avg_ms, 4), "launches_per_step": launches,
                    "bytes_per_ray": round(fetched * launches / max(1, cst["rays_total"] - cst["primaries_culled"]), 1),
                    "rays_traced_per_launch": int((cst["rays_total"] - cst["primaries_culled"]) / launches),
                    "bytes_fetched_closest_hit_rays": int(b_extend), "bytes_fetched_shadow_rays": int(b_shadow),
                    "gather_ceiling": {"GBs": GATHER64_PEAK_GBS, "frac": round(achieved / GATHER64_PEAK_GBS, 4),
                                       "note": "what the vector memory path of this part returns for a divergent 64-byte fetch per lane out of L2, "
                                               "whatever the hit rate or occupancy (profiles/microbench/gather64.hip): the practical ceiling of a record walk"},
                    "reference_equivalent": {"bytes_per_ray": round(reference_bytes_per_ray(), 1), "rays_answered_per_launch": int(rays_answered / launches),
                                             "bytes_per_launch": int(ref_bytes), "GBs": round(ref_gbs, 1), "frac": round(ref_gbs / peak, 4),
                                             "note": "the same formula on what the REFERENCE's walk of its own tree fetches for the rays this launch answers "
                                                     "(the CPU oracle's counters for this frame, checked by cpu_baseline): a measure of what the second tree, "
                                                     "the certificates and the tiles left out of the launch save, not of bandwidth"},
                    "served_by": served_by() if world == 1 else None,
                    "counted_frame": counted_shape(cst), "register_spills": kernel_spills(),
                    "scene_bytes": int(info["device_bytes"]), "hbm_peak": HBM_PEAK_GBS,
                    "note": ("`achieved` = bytes of the records this kernel's own walk fetched (64 per node record visited and per primitive tested, "
                             "48 per ray traced; counters of a counted render of the same frame) / the average launch duration of the timed steps; "
                             "scene + trees (%.1f MB) fit each XCD's 4 MiB L2, so `peak` is the aggregate L2 bandwidth and `level` says so (`bound` "
                             "is the nearer of the two allowed values); `traffic` = fabric-side bytes per launch from the counter passes; "
                             "the kernel is bound by its waves' dependent chains (fetch -> ~165 instructions -> fetch), not by bytes: see "
                             "`counted_frame` and `valu`" % (info["device_bytes"] / 1e6)) if in_l2 else
                            "scene + BVH exceed the XCDs' L2: `peak` is the HBM bandwidth"}
        insts = profile_counter("pmc_sq", "SQ_INSTS_VALU") if world == 1 else None
        flops_done = algorithmic_flops(cst, shading, len(sd.lights)) / launches
        flops_ref = reference_flops_per_ray() * rays_answered / launches
        wave_cycles, wait_any = profile_counter("pmc_sq", "SQ_WAVE_CYCLES"), profile_counter("pmc_sq", "SQ_WAIT_ANY")
        valu = {"kernel": "k_trace<OUT_CHAIN>", "wave_insts_per_launch": insts,
                "issue_peak_per_s": SIMDS * VALU_ISSUE_PER_SIMD,
                "frac": round(insts / (avg_ms * 1e-3) / (SIMDS * VALU_ISSUE_PER_SIMD), 4) if insts else None,
                "flop_equiv_done_per_launch": int(flops_done),
                "achieved_tflops_equiv": round(flops_done / (avg_ms * 1e-3) / 1e12, 2),
                "peak_tflops": FP32_PEAK_TFLOPS, "frac_flop_equiv": round(flops_done / (avg_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4),
                "reference_equivalent_tflops_equiv": round(flops_ref / (avg_ms * 1e-3) / 1e12, 2),
                "waves_parked_share": round(wait_any / wave_cycles, 4) if wave_cycles and wait_any is not None else None,
                "lane_utilisation_node_steps": round((cst["node_visits"] + cst["node_visits_shadow"]) / max(1, 64 * cst["wave_node_steps"]), 3),
                "lane_utilisation_leaf_steps": round((cst["prim_tests"] + cst["prim_tests_shadow"]) / max(1, 64 * cst["wave_leaf_steps"]), 3),
                "note": "wave_insts_per_launch = SQ_INSTS_VALU of the committed counter pass of this command (profiles/); issue peak = 1 024 SIMDs x "
                        "one wave64 instruction per 2 cycles at 2.4 GHz; flop_equiv_done = 60 per box test, 90 per primitive test, 40 per ray and the shaders' "
                        "share on this kernel's own counts; waves_parked_share = SQ_WAIT_ANY / SQ_WAVE_CYCLES of the same pass"}
        if world == 1 and not args.no_cpu_baseline:
            baseline = cpu_baseline(sd, width, height, SAMPLES, shading, jitter, gpu_frame)
        if world == 1 and not args.no_secondary:
            R["scene"].close()
            secondary = secondary_frames(ra, abi, torch, stream)
            try:
                secondary["headline_two_frames_in_flight"] = frames_in_flight(ra, torch, sd, params, int(rays[0]))
            except Exception as e:      # reported, not hidden
                secondary["headline_two_frames_in_flight"] = {"error": str(e)}
            try:
                predicted = predicted_strong_scaling(ra, abi, tiles, torch, sd, shading, jitter, stream, ms_per_step)
            except Exception as e:
                predicted = {"error": str(e)}
            e2e = end_to_end(ra, abi, R["first_ms"] if R["first_ms"] is not None else 0.0, acc["ms_total"] / args.steps)

    if rank == 0:
        frames = {"strong": f"{WIDTH}x{HEIGHT} split {world} ways", "weak": "%dx%d" % frame_size(world, "weak")}
        out = {
            "metric": "Mrays/s (primary+secondary); ms/frame bunny_tasty 1080p", "value": round(value, 2), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "ms_per_frame": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": DATA,
            "value_traced": round((rays[0] - rays[4]) * args.steps / dt / 1e6, 2),
            "config": {"workload": WORKLOAD if (width, height) == (WIDTH, HEIGHT) else WORKLOAD.replace("1920x1080", f"{width}x{height}"),
                       "frame_by_scaling": frames,
                       "per_gpu": f"{width * height // world} pixels x {SAMPLES} samples",
                       "rays_per_frame": int(rays[0]), "primary": int(rays[1]), "shadow": int(rays[2]), "bounce": int(rays[3]),
                       "primaries_answered_by_tile": int(rays[4]),
                       "value_note": "`value` counts every ray the reference traces for the frame; `primaries_answered_by_tile` of its primaries are answered as "
                                     "misses a whole tile at a time and cost nothing: `value_traced` is the rate of the rays that entered the traversal kernel",
                       "parallelism": f"tiles{world}", "tile": abi.RT_TILE, "jitter": "hash"},
            # HIP events of the timed steps: the frame on the caller's stream, k_trace on the stream it runs on; the kernels
            # behind it are timed one by one only on the instrumented scene (events between kernels cost stream time)
            "device_ms_per_step": dict({k: round(v / args.steps, 4) for k, v in acc.items()},
                                       ms_behind_trace=round((acc["ms_total"] - acc["ms_trace"]) / args.steps, 4),
                                       instrumented_frame={k: round(cst[k], 4) for k in ("ms_light", "ms_combine", "ms_resolve")}),
            "roofline": roofline, "valu": valu, "cpu_baseline": baseline, "secondary": secondary,
            "predicted_strong_scaling": predicted, "end_to_end": e2e,
        }
        if other is not None:
            out[other["scaling"]] = other
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
