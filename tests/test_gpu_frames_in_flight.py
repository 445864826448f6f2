"""Two frames in flight: two replicas of a scene, rendered at the same time from two host threads on two streams.

A scene owns its working buffers, so callers that render concurrently use one scene each (include/raytracer_amd.h).
A frame sequence uses that to keep the GPU busy while a frame's last waves finish: the frames must be the bytes a
single scene renders one after the other, whatever the two launches do to each other's timing.
"""
import os
import threading

import numpy as np
import pytest
import torch

import raytracer_amd as ra
from raytracer_amd import abi

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scenes")


@pytest.mark.parametrize("name,shading,spp", [("bunny", abi.RT_COOK_TORRANCE, 4), ("balls2", abi.RT_PHONG, 1), ("iceBot", abi.RT_PHONG, 1)])
def test_two_threads_two_scenes_same_bytes(name, shading, spp):
    sd = ra.SceneData.load(os.path.join(G, name + ".npz"))
    w, h, rounds = 480, 270, 6
    cams = []
    base = {k: np.array(v, np.float32) for k, v in sd.camera.items()}
    for k in range(2 * rounds):
        c = dict(base)
        c["pos"] = base["pos"] + np.array([0.15 * k, 0.05 * k, 0.0], np.float32)
        cams.append(c)
    # the frames one scene renders one after the other
    want = []
    with ra.Scene(sd) as sc:
        for c in cams:
            sc.set_camera(c["pos"], c["up"], c["right"], c["look_at"])
            f, _ = sc.render(w, h, spp, shading, abi.RT_JITTER_HASH, 1)
            want.append(f.copy())
    # the same frames, even ones by one thread and odd ones by another, at the same time
    p = ra.make_params(w, h, spp, shading, abi.RT_JITTER_HASH, 1)
    scenes = [ra.Scene(sd), ra.Scene(sd)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    bufs = [[torch.zeros(h * w * 4, dtype=torch.uint8, device="cuda") for _ in range(rounds)] for _ in range(2)]
    errors = []

    def work(i):
        try:
            for r in range(rounds):
                c = cams[2 * r + i]
                scenes[i].set_camera(c["pos"], c["up"], c["right"], c["look_at"])
                scenes[i].render_device(p, bufs[i][r].data_ptr(), streams[i].cuda_stream)
        except Exception as e:      # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    for sc in scenes:
        sc.close()
    assert not errors, errors
    for r in range(rounds):
        for i in range(2):
            got = bufs[i][r].cpu().numpy().reshape(h, w, 4)
            assert got.tobytes() == want[2 * r + i].tobytes(), (r, i)
