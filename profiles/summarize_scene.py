#!/usr/bin/env python3
"""Condenses the raw output of profiles/collect_scene.sh into profiles/<tag>_<name>_counters.json: per kernel of the run,
the kernel trace's durations and the per-launch values of each counter pass (mean over the launches and the last
launch, which is the steady one: the scripts render a few frames of one scene), and for the traversal kernel the
fabric-side bytes per launch (FETCH_SIZE doubled, as gfx950 tallies 128-byte requests as 64; WRITE_SIZE as reported;
both in KiB) in profiles/<tag>_<name>_traffic.json."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = ("pmc_sq", "pmc_fetch", "pmc_write", "pmc_l2", "pmc_l1")


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def main():
    tag, name = sys.argv[1], sys.argv[2]
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_{name}")
    out = {"command": "rocprofv3 ... -- " + (sys.argv[3] if len(sys.argv) > 3 else "?") + " (profiles/collect_scene.sh)", "kernels": defaultdict(dict)}
    for p in glob.glob(os.path.join(src, "kt", "**", "*kernel_trace.csv"), recursive=True):
        durs = defaultdict(list)
        with open(p) as f:
            for row in sorted(csv.DictReader(f), key=lambda r: int(r["Dispatch_Id"])):
                if short(row["Kernel_Name"]).startswith("k_"):
                    durs[short(row["Kernel_Name"])].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        for k, d in durs.items():
            out["kernels"][k]["kernel_trace"] = {"calls": len(d), "avg_ns": sum(d) / len(d), "min_ns": min(d), "max_ns": max(d), "last_launch_ns": d[-1],
                                                 "total_ns": sum(d)}
    for pas in PASSES:
        for p in glob.glob(os.path.join(src, pas, "**", "*counter_collection.csv"), recursive=True):
            per = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
            with open(p) as f:
                for row in csv.DictReader(f):
                    if short(row["Kernel_Name"]).startswith("k_"):
                        per[short(row["Kernel_Name"])][row["Counter_Name"]][int(row["Dispatch_Id"])] += float(row["Counter_Value"])
            for k, cs in per.items():
                for c, by_dispatch in cs.items():
                    vals = [by_dispatch[d] for d in sorted(by_dispatch)]
                    out["kernels"][k][c] = {"launches": len(vals), "per_launch_mean": sum(vals) / len(vals), "last_launch": vals[-1]}
    json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_{name}_counters.json"), "w"), indent=1, sort_keys=True)
    # the uncounted traversal kernel: fabric-side bytes of its last (steady) launch
    best = None
    for k, v in out["kernels"].items():
        if k.startswith("k_trace<2, false") and "FETCH_SIZE" in v and (best is None or v["kernel_trace"]["total_ns"] > best[1]["kernel_trace"]["total_ns"]):
            best = (k, v)
    if best:
        k, v = best
        fetch, write = v["FETCH_SIZE"]["last_launch"], v.get("WRITE_SIZE", {}).get("last_launch", 0.0)
        last = lambda c: v.get(c, {}).get("last_launch")
        hit, miss = last("TCC_HIT_sum"), last("TCC_MISS_sum")
        tags = [last("TCP_TAGRAM%d_REQ_sum" % i) for i in range(4)]
        json.dump({"kernel": k, "tag": tag, "frame": name, "fetch_size_kib_last_launch_raw": fetch, "write_size_kib_last_launch_raw": write,
                   "correction": "FETCH_SIZE x2 (gfx950 tallies 128-byte requests as 64 bytes); WRITE_SIZE as reported",
                   "hbm_bytes_per_launch": int(fetch * 1024 * 2 + write * 1024),
                   "launch_ns_under_the_kernel_trace": v["kernel_trace"]["last_launch_ns"],
                   "l2_hits_per_launch": hit, "l2_misses_per_launch": miss, "l2_hit_rate": (hit / (hit + miss) if hit is not None and miss is not None and hit + miss > 0 else None),
                   "l1_read_requests_to_l2_per_launch": last("TCP_TCC_READ_REQ_sum"),
                   "l1_tag_requests_per_launch": (sum(tags) if all(t is not None for t in tags) else None)},
                  open(os.path.join(ROOT, "profiles", f"{tag}_{name}_traffic.json"), "w"), indent=1)
    for k, v in out["kernels"].items():
        print(k, {c: round(x.get("last_launch", x.get("last_launch_ns", 0))) for c, x in v.items()})


if __name__ == "__main__":
    main()
