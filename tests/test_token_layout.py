"""include/raytracer_tokens.h against the reference's own TokenData.h (TokenData.h:11-110, with the GLM 0.9.3 the
reference vendors): size and offset of every member of every TK* struct, and the element order of vec3 / mat4.
Runs only where the reference checkout exists (the build container)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"

pytestmark = pytest.mark.ref


def listing(tmp_path, name, flags):
    exe = tmp_path / name
    subprocess.run(["g++", "-std=c++11", "-w", "-Wno-invalid-offsetof", *flags, os.path.join(ROOT, "tests", "native", "token_layout.cpp"),
                    "-o", str(exe)], check=True)
    return subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines()


def test_token_structs_have_the_reference_layout(tmp_path):
    if not os.path.exists(os.path.join(REFERENCE, "TokenData.h")):
        pytest.skip("the reference checkout is not here")
    ours = listing(tmp_path, "ours", ["-I", os.path.join(ROOT, "include")])
    theirs = listing(tmp_path, "theirs", ["-DUSE_REFERENCE_HEADER", "-I", REFERENCE])
    assert len(ours) > 60
    assert ours == theirs
