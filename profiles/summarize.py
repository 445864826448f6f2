#!/usr/bin/env python3
"""Condenses the raw rocprofv3 output of profiles/collect.sh into the small files kept under profiles/.

  <tag>_bench.json            the bench.py line of the same run
  <tag>_kernel_stats.csv      rocprofv3 --kernel-trace --stats: per-kernel calls / total / average ns (all launches of
                              the run: warm-up frames, with the first frame's growth and the second-tree try-out, included)
  <tag>_kernel_timed.json     the dominant kernel's launches of the timed steps alone (the last `steps` of the trace):
                              average / min / max duration, to set beside the bench line's roofline.avg_launch_ms
  <tag>_pmc_summary.json      per-kernel sums and per-launch averages of the counters of each --pmc pass
  traffic_trace_kernel.json   HBM-side bytes per launch of the dominant kernel (the persistent
                              traversal kernel): FETCH_SIZE doubled as the gfx950 note on that counter says
                              (it tallies 128-byte requests as 64), plus WRITE_SIZE as reported; bench.py
                              reads it into roofline.traffic
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOMINANT = "k_trace<2, false"        # the uncounted whole-forest traversal kernel (with or without the stack-spill variant)


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def pmc_table(path):
    """{kernel: {counter: [sum, launches]}} from one counter_collection.csv (one row per dispatch x counter)."""
    out = defaultdict(lambda: defaultdict(lambda: [0.0, 0, []]))
    with open(path) as f:
        for row in csv.DictReader(f):
            cell = out[short(row["Kernel_Name"])][row["Counter_Name"]]
            cell[0] += float(row["Counter_Value"])
            cell[1] += 1
            cell[2].append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
    return out


def timed_mean(pairs, steps):
    """Mean over the last `steps` dispatches (the bench's timed steps; the ones before are its warm-up frames)."""
    per = defaultdict(float)
    for d, v in pairs:
        per[d] += v
    vals = [per[d] for d in sorted(per)][-steps:]
    return sum(vals) / max(1, len(vals))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    if os.path.exists(os.path.join(src, "bench.json")):
        shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, f"{tag}_bench.json"))
    for p in glob.glob(os.path.join(src, "kt", "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(p, os.path.join(dst, f"{tag}_kernel_stats.csv"))
    steps = 20
    if os.path.exists(os.path.join(src, "bench.json")):
        try:
            steps = int(json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])["steps"])
        except (ValueError, KeyError, IndexError):
            pass
    for p in glob.glob(os.path.join(src, "kt", "**", "*kernel_trace.csv"), recursive=True):
        durs = []
        with open(p) as f:
            for row in csv.DictReader(f):
                if DOMINANT in row["Kernel_Name"]:
                    durs.append((int(row["Dispatch_Id"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
        timed = [d for _, d in sorted(durs)][-steps:]
        if timed:
            json.dump({"kernel": DOMINANT, "launches_in_trace": len(durs), "timed_launches": len(timed),
                       "avg_ns": sum(timed) / len(timed), "min_ns": min(timed), "max_ns": max(timed),
                       "note": "the last `timed_launches` launches of the kernel in the --kernel-trace run: the bench's timed steps"},
                      open(os.path.join(dst, f"{tag}_kernel_timed.json"), "w"), indent=1)
    summary = {}
    for pas in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_l2", "pmc_l1"):
        files = glob.glob(os.path.join(src, pas, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            continue
        tab = pmc_table(files[0])
        summary[pas] = {k: {c: {"sum": v[0], "launches": v[1], "per_launch": v[0] / max(1, v[1]),
                                "per_launch_timed_steps": timed_mean(v[2], steps)} for c, v in cs.items()}
                        for k, cs in tab.items() if k.startswith("k_")}
    json.dump(summary, open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w"), indent=1, sort_keys=True)
    def dominant(pas, counter):
        """The counter of the variant of the dominant kernel that the pass launched most often (the timed steps' one)."""
        best = None
        for k, cs in summary.get(pas, {}).items():
            if DOMINANT in k and counter in cs and (best is None or cs[counter]["launches"] > best["launches"]):
                best = cs[counter]
        return best["per_launch_timed_steps"] if best else None
    fetch, write = dominant("pmc_fetch", "FETCH_SIZE"), dominant("pmc_write", "WRITE_SIZE")
    if fetch is not None:
        # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB
        fetch_b = fetch * 1024 * 2
        write_b = (write or 0.0) * 1024
        json.dump({"kernel": DOMINANT, "tag": tag,
                   "fetch_size_kib_per_launch_raw": fetch, "write_size_kib_per_launch_raw": write,
                   "correction": "FETCH_SIZE x2 (gfx950 tallies 128-byte requests as 64 bytes); WRITE_SIZE as reported; "
                                 "16-byte-per-lane gathers of 64-byte records are not the calibrated streaming pattern, "
                                 "so the absolute figure is indicative",
                   "hbm_bytes_per_launch": int(fetch_b + write_b)},
                  open(os.path.join(dst, "traffic_trace_kernel.json"), "w"), indent=1)
    print(open(os.path.join(dst, f"{tag}_bench.json")).read() if os.path.exists(os.path.join(dst, f"{tag}_bench.json")) else "no bench line")


if __name__ == "__main__":
    main()
