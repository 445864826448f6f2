"""Ad-hoc (not a test): frame times of the larger benchmark configurations' stand-ins on one GPU."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra
from raytracer_amd import abi, synth

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "scenes")


def run(label, sd, w, h, spp, shading, fast=False, frames=8):
    t0 = time.time()
    with ra.Scene(sd, fast_bvh=fast) as sc:
        info = sc.info()
        best = None
        for i in range(frames):
            _, st = sc.render(w, h, spp, shading, abi.RT_JITTER_HASH)
            if (i >= 5 or frames < 6 and (i or frames == 1)) and (best is None or st["ms_total"] < best["ms_total"]):
                best = st
    st = best
    print(f"{label}: {w}x{h} spp{spp} {'fast-tree ' if fast else ''}build {info['ms_build_host']:.0f} ms, depth {info['tree_depth']}, frame {st['ms_total']:.2f} ms, "
          f"{st['rays_total'] / 1e6:.1f} Mrays -> {st['rays_total'] / st['ms_total'] / 1e3:.0f} Mrays/s, chunks {st['launches_trace']}, "
          f"attempts {st['attempts']}, second tree {st['second_tree']}  (wall {time.time() - t0:.1f} s)", flush=True)


which = sys.argv[1:] or ["bunny", "glassy", "icebot"]
if "bunny" in which:
    run("config 3 stand-in A (bunny, Cook-Torrance, 4 strata)", ra.SceneData.load(os.path.join(G, "bunny.npz")), 1920, 1080, 4, abi.RT_COOK_TORRANCE)
    run("config 3 stand-in B (bunny.obj, smooth triangles, Cook-Torrance, 4 strata)", ra.SceneData.load(os.path.join(G, "bunny_obj.npz")),
        1920, 1080, 4, abi.RT_COOK_TORRANCE)
if "glassy" in which:
    run("config 4 stand-in (glassy dragon, depth 6)", synth.glassy(ra.SceneData.load(os.path.join(G, "dragon_large_tris.npz"))), 3840, 2160, 1, abi.RT_PHONG)
if "icebot" in which:
    run("iceBot 4K", ra.SceneData.load(os.path.join(G, "iceBot.npz")), 3840, 2160, 1, abi.RT_PHONG)
if "big1" in which or "big16" in which:
    big = synth.terrain(1000, 500)
    if "big1" in which:
        run("config 5 (1M triangles, 256 lights), fast tree, 1 spp", big, 3840, 2160, 1, abi.RT_PHONG, fast=True)
    if "big16" in which:
        run("config 5 (1M triangles, 256 lights), fast tree, 16 spp", big, 3840, 2160, 16, abi.RT_PHONG, fast=True, frames=2)
