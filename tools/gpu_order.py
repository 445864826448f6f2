"""Ad-hoc (not a test): balls2 / bunny frame times over the tile-order parameters."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra
from raytracer_amd import abi
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "scenes")
CASES = (("balls2", 1, abi.RT_PHONG), ("bunny", 4, abi.RT_COOK_TORRANCE), ("dragon_large_tris", 1, abi.RT_PHONG))
scenes = {n: ra.SceneData.load(os.path.join(G, n + ".npz")) for n, _, _ in CASES}
for classes in (64, 256, 1024):
    for weight in (0, 1, 2, 4, 8):
        os.environ["RT_ORDER_CLASSES"] = str(classes); os.environ["RT_ORDER_MEAN_WEIGHT"] = str(weight)
        row = []
        for name, spp, sh in CASES:
            with ra.Scene(scenes[name]) as sc:
                ts = []
                for i in range(12):
                    _, st = sc.render(1920, 1080, spp, sh, abi.RT_JITTER_HASH)
                    if i >= 5:
                        ts.append(st["ms_trace"])
            ts.sort()
            row.append(f"{name} {ts[0]:.2f}/{ts[len(ts)//2]:.2f}")
        print(f"classes {classes} mean weight {weight}:", " ".join(row), flush=True)
