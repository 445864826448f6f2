"""launch_kernel as a caller written against the reference's interface sees it (kernel.h:30-31).

tests/dropin/dropin_main is a C++ program in the shape of the reference's driver: it fills a
TKSceneData, calls launch_kernel and dumps the frame.  It is built in-tree by __graft_entry__.build()
(plain g++, links libraytracer_amd.so).  The same scene is stated below as a SceneData; the frames must
be the same bytes, and both must match the oracle.
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
EXE = os.path.join(ROOT, "tests", "dropin", "dropin_main")


def dropin_scene():
    sd = ra.SceneData()
    sd.camera = {"pos": [0, 1, 9], "up": [0, 1, 0], "right": [1.33333, 0, 0], "look_at": [0, 0.5, 0]}
    sd.lights = np.zeros(2, abi.light_dtype)
    sd.lights["pos"], sd.lights["clr"] = [[-6, 8, 10], [7, 5, 4]], [[1.2, 1.2, 1.2], [0.4, 0.3, 0.3]]

    def mod(clr, filt, fin=None):
        m = ra.identity_modifier()
        m["mat"]["clr"], m["mat"]["alpha"] = clr, filt
        if fin is not None:
            for k, v in zip(("amb", "dif", "spec", "rough", "refl", "refr", "ior"), fin):
                m["mat"][k] = v
        return m

    sd.spheres = np.zeros(2, abi.sphere_dtype)
    sd.spheres["c"], sd.spheres["r"] = [[-1.5, 0.5, 0], [2.0, 0.2, 1.5]], [1.5, 1.0]
    sd.spheres["mod"][0] = mod([0.9, 0.2, 0.2], 0.0, (0.2, 0.5, 0.6, 0.05, 0.4, 0.0, 1.0))
    sd.spheres["mod"][1] = mod([0.9, 0.9, 1.0], 0.7, (0.1, 0.3, 0.5, 0.02, 0.0, 1.0, 1.33))
    sd.planes = np.zeros(1, abi.plane_dtype)
    sd.planes["n"], sd.planes["d"] = [[0, 1, 0]], [-1.0]
    sd.planes["mod"][0] = mod([0.3, 0.5, 0.8], 0.0, (0.3, 0.6, 0.0, 0.05, 0.2, 0.0, 1.0))
    sd.triangles = np.zeros(1, abi.triangle_dtype)
    sd.triangles["p1"], sd.triangles["p2"], sd.triangles["p3"] = [[-4, -1, -3]], [[4, -1, -3]], [[0, 5, -4]]
    sd.triangles["mod"][0] = mod([0.2, 0.8, 0.3], 0.0)
    sd.boxes = np.zeros(1, abi.box_dtype)
    sd.boxes["p1"], sd.boxes["p2"] = [[0.2, -1.0, -1.5]], [[1.4, 0.4, -0.4]]
    sd.boxes["mod"][0] = mod([0.8, 0.7, 0.1], 0.0, (0.2, 0.7, 0.3, 0.1, 0.0, 0.0, 1.0))
    return sd


def run_dropin(tmp_path, w, h, samples, shading, jitter_env=None):
    if not os.path.exists(EXE):
        pytest.fail(f"{EXE} is missing: __graft_entry__.build() makes it")
    out = str(tmp_path / "frame.rgba")
    env = dict(os.environ)
    if jitter_env is not None:
        env["RT_JITTER"] = jitter_env
    r = subprocess.run([EXE, out, str(w), str(h), str(samples), shading], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    return np.fromfile(out, np.uint8).reshape(h, w, 4), r.stdout


@pytest.mark.parametrize("shading,samples", [("phong", 1), ("cook", 1), ("phong", 4)])
def test_dropin_frame_equals_c_abi_frame(tmp_path, shading, samples):
    w, h = 200, 150
    got, _ = run_dropin(tmp_path, w, h, samples, shading)            # hash jitter on, like the reference's default
    sd = dropin_scene()
    st = abi.RT_COOK_TORRANCE if shading == "cook" else abi.RT_PHONG
    with ra.Scene(sd) as sc:
        want, _ = sc.render(w, h, samples, st, abi.RT_JITTER_HASH)
    assert got.tobytes() == want.tobytes()
    osc = orclib.OracleScene(sd)
    ref = osc.render(w, h, samples, st, abi.RT_JITTER_HASH, want_samples=False, want_primary=False)["rgba"]
    osc.close()
    diff = np.abs(got[:, :, :3].astype(int) - ref[:, :, :3].astype(int)).max(axis=2)
    assert (diff <= 1).mean() >= 0.999
    assert len(np.unique(got[:, :, :3].reshape(-1, 3), axis=0)) > 100          # a picture, not a constant


def test_dropin_jitter_switch_and_bad_sample_count(tmp_path):
    w, h = 64, 48
    got, _ = run_dropin(tmp_path, w, h, 1, "phong", jitter_env="0")
    with ra.Scene(dropin_scene()) as sc:
        want, _ = sc.render(w, h, 1, abi.RT_PHONG, abi.RT_JITTER_NONE)
    assert got.tobytes() == want.tobytes()
    frame, out = run_dropin(tmp_path, w, h, 5, "phong")                         # kernel.cu:707-712
    assert "Invalid sample count: 5" in out
    assert (frame == 7).all()                                                   # the caller's buffer is untouched
