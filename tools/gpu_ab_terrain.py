"""Ad-hoc (not a test): the 1M-triangle / 256-light scene at 1080p 4 spp, best of 3 frames (for A/B runs with RT_LIBRARY)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra
from raytracer_amd import abi, synth
sd = synth.terrain(1000, 500, lights_per_side=16)
with ra.Scene(sd, fast_bvh=True) as sc:
    ts = []
    for i in range(4):
        _, st = sc.render(1920, 1080, 4, abi.RT_PHONG, abi.RT_JITTER_HASH)
        ts.append(st["ms_total"])
print(os.environ.get("RT_LIBRARY", "default"), "terrain 1080p 4spp ms:", " ".join(f"{t:.1f}" for t in ts), "; feelers", st["rays_shadow"], "of them stopped after their own-surface test", st["shadows_unneeded"], flush=True)
