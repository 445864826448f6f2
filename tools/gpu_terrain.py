"""Ad-hoc (not a test): per-kernel times of the 1M-triangle / 256-light scene at a reduced frame."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra
from raytracer_amd import abi, synth
sd = synth.terrain(1000, 500, lights_per_side=16)
w, h, spp = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (1920, 1080, 4)))
with ra.Scene(sd, fast_bvh=True) as sc:
    for i in range(3):
        _, st = sc.render(w, h, spp, abi.RT_PHONG, abi.RT_JITTER_HASH)
        print(f"fuse={os.environ.get('RT_FUSE_RESOLVE','1')} frame {i}: total {st['ms_total']:.1f} = trace {st['ms_trace']:.1f} + light {st['ms_light']:.1f} + combine {st['ms_combine']:.1f} + resolve {st['ms_resolve']:.1f}; "
              f"rays {st['rays_total']/1e6:.0f} M, chunks {st['launches_trace']}", flush=True)
