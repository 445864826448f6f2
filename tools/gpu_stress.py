"""Ad-hoc (not a test): many random scenes and views, every shortcut on against every shortcut off.

For each seed a random scene (every primitive class, random finishes, with and without a plane, 1-5 lights), a random
camera and frame size are rendered twice: with the defaults (second tree, tiles left out, feelers that cannot matter
stopped early, costliest tiles first) and with all of that switched off (RT_MISS_FILTER=0 RT_CULL_TILES=0
RT_SKIP_UNLIT=0 RT_TILE_ORDER=0: every ray makes the reference's walk).  Frames must be the same bytes and the ray
counts equal.  python tools/gpu_stress.py [first_seed] [count]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra
from raytracer_amd import abi, synth

OFF = {"RT_MISS_FILTER": "0", "RT_CULL_TILES": "0", "RT_SKIP_UNLIT": "0", "RT_TILE_ORDER": "0"}
COUNTS = ("rays_total", "rays_primary", "rays_shadow", "rays_reflect", "rays_refract", "shade_nodes", "shadow_skipped")


def render(sd, env, w, h, spp, shading, frames):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        with ra.Scene(sd) as sc:
            for _ in range(frames):
                f, st = sc.render(w, h, spp, shading, abi.RT_JITTER_HASH, 3)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return f, st


first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad = 0
culled = unneeded = 0
for seed in range(first, first + count):
    rng = np.random.Generator(np.random.PCG64(10_000 + seed))
    sd = synth.random_scene(seed, n_spheres=int(rng.integers(0, 40)), n_triangles=int(rng.integers(0, 60)), n_boxes=int(rng.integers(0, 12)),
                            n_smooth=int(rng.integers(0, 40)), n_planes=int(rng.integers(0, 2)), n_lights=int(rng.integers(1, 6)))
    if len(sd.spheres) + len(sd.triangles) + len(sd.boxes) + len(sd.smooth_triangles) == 0:
        continue
    pos = rng.normal(size=3) * rng.choice([1.0, 6.0, 20.0])
    at = rng.normal(size=3) * rng.choice([0.5, 3.0])
    sd.camera = {"pos": pos.astype(np.float32), "up": np.array([0, 1, 0], np.float32),
                 "right": np.array([rng.choice([1.0, 1.3333, 1.7778]), 0, 0], np.float32), "look_at": at.astype(np.float32)}
    w, h = int(rng.integers(40, 400)), int(rng.integers(40, 300))
    spp = int(rng.choice([1, 1, 4, 9]))
    shading = abi.RT_PHONG if rng.random() < 0.6 else abi.RT_COOK_TORRANCE
    frames = int(rng.choice([1, 2, 3]))
    os.environ["RT_FILTER_MIN_PRIMS"] = "4"          # small scenes get a second tree too
    f1, s1 = render(sd, {}, w, h, spp, shading, frames)
    f0, s0 = render(sd, OFF, w, h, spp, shading, 1)
    ok = f1.tobytes() == f0.tobytes() and all(s1[k] == s0[k] for k in COUNTS)
    culled += s1["primaries_culled"]; unneeded += s1["shadows_unneeded"]
    if not ok:
        bad += 1
        diff = int((f1 != f0).any(axis=-1).sum())
        print(f"seed {seed}: MISMATCH {w}x{h} spp {spp} shading {shading} frames {frames}: {diff} pixels differ; "
              + " ".join(f"{k} {s1[k]}/{s0[k]}" for k in COUNTS if s1[k] != s0[k]), flush=True)
    if (seed - first) % 25 == 24:
        print(f"... {seed - first + 1} scenes, {bad} mismatches, {culled} primaries answered by tile, {unneeded} feelers stopped early", flush=True)
print(f"{count} scenes from seed {first}: {bad} mismatches", flush=True)
sys.exit(1 if bad else 0)
