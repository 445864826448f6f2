#!/usr/bin/env python3
"""Registers, spills and scratch of every kernel of a HIP source, from hipcc's own resource remarks.

    python tools/kernel_resources.py [source.hip ...] > profiles/rNN_kernel_resources.txt

Compiles the device side only (no GPU needed) with the flags of raytracer_amd/csrc/Makefile plus
-Rpass-analysis=kernel-resource-usage and prints one line per kernel."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "raytracer_amd", "csrc")
FIELDS = ("VGPRs", "AGPRs", "VGPRs Spill", "SGPRs Spill", "TotalSGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]")


def resources(src, extra=()):
    # the library's own flags (csrc/Makefile): -O3, and -Os on top of it for the translation unit with the kernels
    level = ["-O3", "-Os"] if os.path.basename(src) == "rt_render.hip" else ["-O3"]
    cmd = ["/opt/rocm/bin/hipcc", *level, "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-I" + os.path.join(ROOT, "include"),
           "-I" + CSRC, "--offload-arch=gfx950", "-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", *extra, src, "-o", "/dev/null"]
    err = subprocess.run(cmd, capture_output=True, text=True, cwd=CSRC).stderr
    out, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark: \s*([^:]+): (.*?) \[-Rpass", line)
        if not m:
            continue
        key, val = m.group(1).strip(), m.group(2).strip()
        if key == "Function Name":
            cur = {"name": val}
            out.append(cur)
        elif cur is not None:
            cur[key] = val
    return out


def demangle(names):
    try:
        r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
        out = r.stdout.splitlines()
        return out if len(out) == len(names) else names
    except OSError:
        return names


def main():
    srcs = [os.path.abspath(a) for a in sys.argv[1:]] or [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]
    for src in srcs:
        ks = resources(src)
        names = demangle([k["name"] for k in ks])
        print(f"# {os.path.relpath(src, ROOT)}")
        print("# " + " | ".join(("kernel",) + FIELDS))
        for k, n in zip(ks, names):
            n = re.sub(r"\(.*", "", n)
            print(" | ".join([n] + [str(k.get(f, "-")) for f in FIELDS]))


if __name__ == "__main__":
    main()
