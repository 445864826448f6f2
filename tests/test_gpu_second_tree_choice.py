"""Which rays start in the second tree is tried out per scene and frame size, and never changes a byte.

The first frames of a size go through the four combinations (closest-hit rays x shadow feelers), the quickest stays;
RT_MISS_FILTER=1 pins "both".  Every frame of the series must be the same bytes, and rt_stats.second_tree must say
what was used.
"""
import os

import pytest

import raytracer_amd as ra
from raytracer_amd import abi

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scenes")


def series(sd, n, w=320, h=200, env=None):
    old = os.environ.get("RT_MISS_FILTER")
    if env is None:
        os.environ.pop("RT_MISS_FILTER", None)
    else:
        os.environ["RT_MISS_FILTER"] = env
    try:
        with ra.Scene(sd) as sc:
            out = []
            for _ in range(n):
                frame, st = sc.render(w, h, 1, abi.RT_PHONG, abi.RT_JITTER_HASH, 2)
                out.append((frame.tobytes(), st["second_tree"], st["rays_total"]))
            frame2, st2 = sc.render(w + 32, h, 1, abi.RT_PHONG, abi.RT_JITTER_HASH, 2)      # another size: the try-out starts again
            return out, st2["second_tree"]
    finally:
        if old is None:
            os.environ.pop("RT_MISS_FILTER", None)
        else:
            os.environ["RT_MISS_FILTER"] = old


@pytest.mark.parametrize("name", ["iceBot", "balls2", "bunny"])
def test_try_out_goes_through_the_combinations_and_settles(name):
    sd = ra.SceneData.load(os.path.join(G, name + ".npz"))
    out, after_resize = series(sd, 9)
    assert len({o[0] for o in out}) == 1 and len({o[2] for o in out}) == 1
    kinds = [o[1] for o in out]
    assert kinds[0] == (0 if name == "iceBot" else 3) and kinds[1:5] == [3, 1, 2, 0]      # iceBot is mostly glass: its first guess is "neither"
    assert len(set(kinds[5:])) == 1 and kinds[5] in (0, 1, 2, 3)
    assert after_resize == kinds[5]           # the first frame of another size runs on the choice so far


def test_pinned_by_the_environment():
    sd = ra.SceneData.load(os.path.join(G, "iceBot.npz"))
    out, after_resize = series(sd, 7, env="1")
    assert [o[1] for o in out] == [3] * 7 and after_resize == 3
    assert len({o[0] for o in out}) == 1
