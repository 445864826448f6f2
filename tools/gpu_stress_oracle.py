"""Ad-hoc (not a test): random scenes and views against the CPU oracle -- primary hit ids and distances bit for bit,
the ray bookkeeping, 8-bit pixels within 1.  python tools/gpu_stress_oracle.py [first_seed] [count]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orclib
import raytracer_amd as ra
from raytracer_amd import abi, synth

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
os.environ["RT_FILTER_MIN_PRIMS"] = "4"
bad = 0
for seed in range(first, first + count):
    rng = np.random.Generator(np.random.PCG64(20_000 + seed))
    sd = synth.random_scene(seed, n_spheres=int(rng.integers(0, 40)), n_triangles=int(rng.integers(0, 60)), n_boxes=int(rng.integers(0, 12)),
                            n_smooth=int(rng.integers(0, 40)), n_planes=int(rng.integers(0, 2)), n_lights=int(rng.integers(1, 6)))
    if len(sd.spheres) + len(sd.triangles) + len(sd.boxes) + len(sd.smooth_triangles) == 0:
        continue
    pos = rng.normal(size=3) * rng.choice([1.0, 6.0, 20.0])
    at = rng.normal(size=3) * rng.choice([0.5, 3.0])
    sd.camera = {"pos": pos.astype(np.float32), "up": np.array([0, 1, 0], np.float32),
                 "right": np.array([rng.choice([1.0, 1.3333, 1.7778]), 0, 0], np.float32), "look_at": at.astype(np.float32)}
    w, h = int(rng.integers(40, 240)), int(rng.integers(40, 180))
    spp = int(rng.choice([1, 1, 4]))
    shading = abi.RT_PHONG if rng.random() < 0.6 else abi.RT_COOK_TORRANCE
    ref = orclib.OracleScene(sd).render(w, h, spp, shading, abi.RT_JITTER_HASH, 3)
    with ra.Scene(sd) as sc:
        prim, t = sc.trace_primary(w, h, spp, abi.RT_JITTER_HASH, 3)
        frame, st = sc.render(w, h, spp, shading, abi.RT_JITTER_HASH, 3)
        frame, st = sc.render(w, h, spp, shading, abi.RT_JITTER_HASH, 3)
    c = ref["counters"]
    diff = np.abs(frame[:, :, :3].astype(int) - ref["rgba"][:, :, :3].astype(int)).max(axis=2)
    problems = []
    if not np.array_equal(prim, ref["prim"]): problems.append(f"{(prim != ref['prim']).sum()} ids")
    if not np.array_equal(t.view(np.uint32), ref["t"].view(np.uint32)): problems.append("t")
    for a, b in (("rays_primary", "rays_primary"), ("rays_reflect", "rays_reflect"), ("rays_refract", "rays_refract"), ("shade_nodes", "shade_nodes"),
                 ("shadow_skipped", "shadow_selfmiss")):
        if st[a] != c[b]: problems.append(f"{a} {st[a]}/{c[b]}")
    if st["rays_shadow"] + st["shadow_skipped"] != c["rays_shadow"]: problems.append("rays_shadow")
    if (diff <= 1).mean() < 0.999: problems.append(f"{(diff > 1).sum()} pixels off by more than 1")
    if problems:
        bad += 1
        print(f"seed {seed} ({w}x{h} spp {spp} shading {shading}):", ", ".join(problems), flush=True)
    if (seed - first) % 25 == 24:
        print(f"... {seed - first + 1} scenes, {bad} with differences", flush=True)
print(f"{count} scenes from seed {first}: {bad} with differences", flush=True)
sys.exit(1 if bad else 0)
