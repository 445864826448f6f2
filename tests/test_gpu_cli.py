"""The command-line driver end to end (the reference's raytrace.cpp:18-120): parse a .pov, render on the GPU, write a TGA.

`raytracer_amd/raytrace` is built by __graft_entry__.build().  Its file must be, byte for byte, what rt_write_tga makes
of the frame the C ABI renders for the same scene, and that frame must match the oracle's (the 8-bit bar of the other
parity tests: device powf / expf against glibc's).  Also covered: the anti-aliasing and shading flags, the reference's
message for a sample count without an integer square root (kernel.cu:707-712), and the usage errors.
"""
import os
import subprocess

import numpy as np
import pytest

import orclib
import raytracer_amd as ra
from raytracer_amd import abi

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "raytracer_amd", "raytrace")
POV = os.path.join(ROOT, "tests", "golden", "pov", "grammar_mix.pov")


def run(args, cwd):
    if not os.path.exists(EXE):
        pytest.fail(f"{EXE} is missing: __graft_entry__.build() makes it")
    return subprocess.run([EXE, *args], cwd=cwd, capture_output=True, text=True, timeout=120)


def read_tga(path):
    raw = open(path, "rb").read()
    w, h = raw[12] | (raw[13] << 8), raw[14] | (raw[15] << 8)
    assert raw[2] == 2 and raw[16] == 24 and len(raw) == 18 + 3 * w * h        # uncompressed true colour, 24 bits
    return raw, np.frombuffer(raw, np.uint8, offset=18).reshape(h, w, 3)[:, :, ::-1]     # BGR -> RGB, rows bottom -> top


@pytest.mark.parametrize("flags, samples, shading", [([], 1, abi.RT_PHONG), (["-p"], 1, abi.RT_PHONG), (["-t", "-s", "4"], 4, abi.RT_COOK_TORRANCE)])
def test_cli_writes_the_frame_of_the_c_abi(tmp_path, flags, samples, shading):
    w, h = 160, 120
    out = tmp_path / "out.tga"
    r = run([str(w), str(h), "-j0", *flags, "-I", POV, "-O", str(out)], tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    raw, rgb = read_tga(out)

    sd = ra.parse_pov(POV)
    with ra.Scene(sd) as sc:
        frame, _ = sc.render(w, h, samples, shading, abi.RT_JITTER_NONE)
    same = tmp_path / "abi.tga"
    ra.write_tga(same, frame)
    assert open(same, "rb").read() == raw                      # the CLI is the C ABI plus the TGA writer, nothing else

    osc = orclib.OracleScene(sd)
    try:
        ref = osc.render(w, h, samples, shading, abi.RT_JITTER_NONE, want_samples=False, want_primary=False)["rgba"]
    finally:
        osc.close()
    want = tmp_path / "oracle.tga"
    ra.write_tga(want, ref)
    _, rgb_ref = read_tga(want)
    diff = np.abs(rgb.astype(int) - rgb_ref.astype(int)).max(axis=2)
    assert (diff <= 1).mean() >= 0.999
    assert len(np.unique(rgb.reshape(-1, 3), axis=0)) > 100        # a picture, not a constant


def test_cli_default_output_name_and_size(tmp_path):
    r = run(["-j0", "-I" + POV], tmp_path)                        # -Ifile without a space, default 800x600 -> sample.tga
    assert r.returncode == 0, r.stdout + r.stderr
    raw, rgb = read_tga(tmp_path / "sample.tga")
    assert rgb.shape == (600, 800, 3)


def test_cli_rejects_a_sample_count_without_integer_root(tmp_path):
    out = tmp_path / "none.tga"
    r = run(["64", "48", "-s", "3", "-I", POV, "-O", str(out)], tmp_path)
    assert "Invalid sample count: 3. Sample count for anti aliasing must have an integer square root" in r.stdout
    assert not out.exists()


def test_cli_usage_errors(tmp_path):
    r = run(["64", "48"], tmp_path)
    assert r.returncode != 0 and "must specify an input file" in r.stdout
    r = run(["64", "48", "-I", str(tmp_path / "missing.pov")], tmp_path)
    assert r.returncode != 0 and "Error parsing file" in r.stdout
    r = run(["--help"], tmp_path)
    assert r.returncode == 0 and "-p (phong) | -t (cook-torrance)" in r.stdout
