"""Ad-hoc (not a test): frame times over a grid of traversal tuning knobs (read from the environment per scene)."""
import itertools, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra
from raytracer_amd import abi
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "scenes")
CASES = (("bunny", 4, abi.RT_COOK_TORRANCE), ("balls2", 1, abi.RT_PHONG), ("dragon_large_tris", 1, abi.RT_PHONG))
scenes = {n: ra.SceneData.load(os.path.join(G, n + ".npz")) for n, _, _ in CASES}
grid = {"RT_LEAF_TRIGGER": (24, 32, 40), "RT_NODE_BURST": (8, 12, 16), "RT_REFILL_IDLE": (16, 24, 32)}
keys = list(grid)
for combo in itertools.product(*[grid[k] for k in keys]):
    for k, v in zip(keys, combo):
        os.environ[k] = str(v)
    row = []
    for name, spp, shading in CASES:
        with ra.Scene(scenes[name]) as sc:
            ts = []
            for i in range(9):
                _, st = sc.render(1920, 1080, spp, shading, abi.RT_JITTER_HASH)
                if i >= 3:
                    ts.append(st["ms_total"])
        ts.sort()
        row.append(f"{name} {ts[0]:.3f}/{ts[len(ts)//2]:.3f}")
    print(dict(zip(keys, combo)), " ".join(row), flush=True)
