import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, raytracer_amd as ra
from raytracer_amd import abi
sd = ra.SceneData.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scenes", "box.npz"))
with ra.Scene(sd) as sc:
    for i in range(6):
        t0 = time.time(); prim, t = sc.trace_primary(160, 120); print("primary", time.time() - t0, flush=True)
        t0 = time.time(); rgb, st = sc.render_samples(160, 120, 1, 0, 0); print("samples", time.time() - t0, st['rays_total'], flush=True)
        t0 = time.time(); fr, st = sc.render(160, 120, 1, 0, 0); print("render", time.time() - t0, st['rays_total'], flush=True)
