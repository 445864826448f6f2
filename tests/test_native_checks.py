"""CPU checks of the two arithmetic shortcuts the HIP traversal relies on (tests/native/*.c).

Both programs compare the shortcut with the reference's literal arithmetic bit for bit:
  exact_div_hard_cases.c  every operand pair whose quotient is close enough to a rounding boundary
                          for the 3-operation division to be able to misround (exhaustive)
  slab_short_form.c       the short box test against boxIntersect (GeometryUtil.h:9-51), randomised, with and
                          without axis-parallel direction components
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
