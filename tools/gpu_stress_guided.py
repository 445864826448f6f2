"""Ad-hoc (not a test): contested hits on purpose.  Random stacks of near-coincident triangle layers (spacings from 0 to
0.06 along the view direction, each layer tilted a little so that the spacing sweeps across the tie window and the guided
walk's limits within one frame), some with spheres poked through them; rendered with the second tree and without it, and
the primary hits, sample colours and frames compared bit for bit.  python tools/gpu_stress_guided.py [first_seed] [count]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as ra
from raytracer_amd import abi, synth

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
os.environ["RT_FILTER_MIN_PRIMS"] = "2"
bad = 0
guided = contested = 0
for seed in range(first, first + count):
    rng = np.random.Generator(np.random.PCG64(50_000 + seed))
    n = int(rng.integers(10, 120))
    layers = int(rng.integers(2, 8))
    sd = ra.SceneData()
    base = ((rng.random((n, 3)) - 0.5) * np.array([6.0, 4.0, 1.0])).astype(np.float32)
    p2 = base + ((rng.random((n, 3)) - 0.5) * np.array([2.0, 2.0, 0.2])).astype(np.float32)
    p3 = base + ((rng.random((n, 3)) - 0.5) * np.array([2.0, 2.0, 0.2])).astype(np.float32)
    tr = np.zeros(n * layers, abi.triangle_dtype)
    mod = synth.identity_modifier(len(tr))
    mod["mat"]["clr"] = rng.random((len(tr), 3)).astype(np.float32)
    mod["mat"]["amb"], mod["mat"]["dif"], mod["mat"]["spec"], mod["mat"]["rough"] = 0.2, 0.6, 0.3, 0.1
    if rng.random() < 0.3:
        mod["mat"]["refl"] = np.where(rng.random(len(tr)) < 0.5, 0.4, 0.0)
    for k in range(layers):
        sl = slice(k * n, (k + 1) * n)
        dz = 0.0 if k == 0 else float(rng.choice([0.0, rng.random() * 0.012, rng.random() * 0.06]))
        tilt = (rng.random(2) - 0.5) * 0.01 if k else np.zeros(2)
        def moved(p):
            q = p.copy()
            q[:, 2] += np.float32(dz) + (p[:, 0] * tilt[0] + p[:, 1] * tilt[1]).astype(np.float32)
            return q
        tr["p1"][sl], tr["p2"][sl], tr["p3"][sl] = moved(base), moved(p2), moved(p3)
    tr["mod"] = mod
    sd.triangles = tr
    if rng.random() < 0.4:
        m = int(rng.integers(1, 12))
        sp = np.zeros(m, abi.sphere_dtype)
        sp["c"] = ((rng.random((m, 3)) - 0.5) * np.array([5.0, 3.0, 0.5])).astype(np.float32)
        sp["r"] = (0.1 + rng.random(m) * 0.6).astype(np.float32)
        smod = synth.identity_modifier(m)
        smod["mat"]["clr"] = rng.random((m, 3)).astype(np.float32)
        smod["mat"]["amb"], smod["mat"]["dif"] = 0.2, 0.7
        sp["mod"] = smod
        sd.spheres = sp
    lt = np.zeros(2, abi.light_dtype)
    lt["pos"], lt["clr"] = [(3.0, 5.0, 9.0), (-4.0, 2.0, 6.0)], (0.7, 0.7, 0.7)
    sd.lights = lt
    cam = np.array([rng.normal() * 1.5, rng.normal() * 1.5, rng.choice([3.0, 8.0, 30.0])])
    sd.camera = {"pos": cam.astype(np.float32), "up": np.array([0, 1, 0], np.float32),
                 "right": np.array([1.3333, 0, 0], np.float32), "look_at": np.array([0, 0, 0], np.float32)}
    w, h = int(rng.integers(100, 320)), int(rng.integers(80, 240))
    spp = int(rng.choice([1, 4]))
    got = []
    for use in (True, False):
        with ra.Scene(sd, miss_filter=use, counters=True) as sc:
            frame, st = sc.render(w, h, spp, abi.RT_PHONG, abi.RT_JITTER_HASH, seed)
            prim, t = sc.trace_primary(w, h, spp, abi.RT_JITTER_HASH, seed)
            rgb, _ = sc.render_samples(w, h, spp, abi.RT_PHONG, abi.RT_JITTER_HASH, seed)
            got.append((frame, prim, t, rgb, st))
    (f1, p1, t1, c1, s1), (f0, p0, t0, c0, s0) = got
    problems = []
    if not np.array_equal(p1, p0): problems.append(f"{(p1 != p0).sum()} ids")
    if not np.array_equal(t1.view(np.uint32), t0.view(np.uint32)): problems.append("t")
    if not np.array_equal(c1.view(np.uint32), c0.view(np.uint32)): problems.append(f"{(c1.view(np.uint32) != c0.view(np.uint32)).any(axis=-1).sum()} sample colours")
    if f1.tobytes() != f0.tobytes(): problems.append("frame")
    guided += s1["hits_guided"]; contested += s1["hits_contested"]
    if problems:
        bad += 1
        print(f"seed {seed} ({w}x{h} spp {spp}, {layers} layers of {n}):", ", ".join(problems), flush=True)
    if (seed - first) % 25 == 24:
        print(f"... {seed - first + 1} scenes, {bad} with differences; contested hits {contested}, of them guided {guided}", flush=True)
print(f"{count} scenes from seed {first}: {bad} with differences; contested hits {contested}, of them guided {guided}", flush=True)
sys.exit(1 if bad else 0)
