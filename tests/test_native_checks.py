"""CPU checks of the shortcuts that stand in for the reference's literal arithmetic (tests/native/*).

Each program compares the shortcut with the literal form bit for bit:
  exact_div_hard_cases.c  every operand pair whose quotient is close enough to a rounding boundary
                          for the 3-operation division to be able to misround (exhaustive)
  slab_short_form.c       the short box test against boxIntersect (GeometryUtil.h:9-51), randomised, with and
                          without axis-parallel direction components
  host_build_sanitized.cpp  the host side of scene construction under ASan + UBSan (no shortcut: a memory-safety run)
  bvh_sort_equivalence.cpp  the radix sort that stands in for formBVH's merge rounds (bvh.cpp:106-171) in the host
                          tree build, against the merge itself: the same permutation on every input
"""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def has_fma():
    try:
        return " fma " in open("/proc/cpuinfo").read()
    except OSError:
        return False


@pytest.mark.skipif(not has_fma(), reason="needs a CPU with FMA to finish in seconds")
def test_exact_division_hard_cases(tmp_path):
    exe = tmp_path / "hard"
    subprocess.run(["gcc", "-O2", "-mfma", "-fopenmp", "-ffp-contract=off",
                    os.path.join(HERE, "native", "exact_div_hard_cases.c"), "-o", str(exe), "-lm"], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    assert "bad=0" in out, out
    tested = int(out.split("tested=")[1].split()[0])
    assert tested > 100_000_000, out


@pytest.mark.skipif(not has_fma(), reason="needs a CPU with FMA to finish in seconds")
def test_short_slab_form(tmp_path):
    exe = tmp_path / "slab"
    subprocess.run(["gcc", "-O2", "-mfma", "-fopenmp", "-ffp-contract=off", "-DTRIALS=20000000LL",
                    os.path.join(HERE, "native", "slab_short_form.c"), "-o", str(exe), "-lm"], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    assert "bad=0" in out and "bitdiff(incl zero sign)=0" in out, out
    assert int(out.split("with_parallel_axis=")[1].split()[0]) > 1_000_000, out


def test_bvh_sort_rounds_radix_equals_merge(tmp_path):
    exe = tmp_path / "bvhsort"
    root = os.path.dirname(HERE)
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(root, "include"), "-I", os.path.join(root, "raytracer_amd", "csrc"),
                    os.path.join(HERE, "native", "bvh_sort_equivalence.cpp"), "-o", str(exe), "-lpthread"], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    assert "bad=0" in out, out
    assert int(out.split("tested=")[1].split()[0]) > 15_000 and int(out.split("with_ties=")[1].split()[0]) > 5_000, out


def test_host_scene_build_under_sanitizers(tmp_path):
    import glob
    exe = tmp_path / "host_asan"
    root = os.path.dirname(HERE)
    csrc = os.path.join(root, "raytracer_amd", "csrc")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
                    "-ffp-contract=off", "-I", os.path.join(root, "include"), "-I", csrc,
                    os.path.join(HERE, "native", "host_build_sanitized.cpp"), os.path.join(csrc, "rt_scene_build.cpp"), os.path.join(csrc, "rt_pov.cpp"),
                    "-o", str(exe), "-lpthread"], check=True)
    scenes = sorted(glob.glob(os.path.join(HERE, "golden", "pov", "*.pov")))
    assert scenes
    out = subprocess.run([str(exe)] + scenes, capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert out.returncode == 0, out.stderr[-4000:]
    assert "ERROR: AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr and "LeakSanitizer" not in out.stderr, out.stderr[-4000:]
    assert "synthetic flags 0: rc 0 prims 70000 nodes 69999 filter 69999 materials 977" in out.stdout, out.stdout[-2000:]
    assert "synthetic deferred: pending 1 same 1" in out.stdout, out.stdout[-2000:]
