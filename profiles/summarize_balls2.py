#!/usr/bin/env python3
"""Condenses the raw output of profiles/collect_balls2.sh into profiles/<tag>_balls2_counters.json: per variant of the
traversal kernel, the kernel trace's durations and the per-launch averages of each counter pass (the uncounted
variant's launches are the six frames of tools/gpu_breakdown.py's first scene: the first sizes the queues, the next four
try out the second tree for both kinds of ray, closest-hit rays, feelers, neither; the sixth runs on the choice made
-- that one is `last_launch`; the counted variant runs twice at the end)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    return name.split("(")[0].strip()


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_balls2")
    out = {"command": "rocprofv3 ... -- python3 tools/gpu_breakdown.py balls2 (profiles/collect_balls2.sh)",
           "frame": "balls2.pov 1920x1080 Phong 1 spp, reference tree", "kernels": defaultdict(dict)}
    for p in glob.glob(os.path.join(src, "kt", "**", "*kernel_trace.csv"), recursive=True):
        durs = defaultdict(list)
        with open(p) as f:
            for row in csv.DictReader(f):
                if "k_trace" in row["Kernel_Name"]:
                    durs[short(row["Kernel_Name"])].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        for k, d in durs.items():
            steady = d[2:] if len(d) > 3 else d          # the first two frames size the queues and try out the second tree
            out["kernels"][k]["kernel_trace"] = {"calls": len(d), "avg_ns_after_the_first_two": sum(steady) / len(steady),
                                                 "min_ns": min(d), "max_ns": max(d), "last_launch": d[-1]}
    for pas in ("pmc_sq", "pmc_fetch", "pmc_write"):
        for p in glob.glob(os.path.join(src, pas, "**", "*counter_collection.csv"), recursive=True):
            per = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
            with open(p) as f:
                for row in csv.DictReader(f):
                    if "k_trace" in row["Kernel_Name"]:
                        per[short(row["Kernel_Name"])][row["Counter_Name"]][int(row["Dispatch_Id"])] += float(row["Counter_Value"])
            for k, cs in per.items():
                for c, by_dispatch in cs.items():
                    vals = [by_dispatch[d] for d in sorted(by_dispatch)]
                    steady = vals[2:] if len(vals) > 3 else vals
                    out["kernels"][k][c] = {"launches": len(vals), "per_launch_after_the_first_two": sum(steady) / len(steady),
                                            "last_launch": vals[-1]}
    json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_balls2_counters.json"), "w"), indent=1, sort_keys=True)
    for k, v in out["kernels"].items():
        print(k, {c: round(x["last_launch"]) for c, x in v.items()})


if __name__ == "__main__":
    main()
