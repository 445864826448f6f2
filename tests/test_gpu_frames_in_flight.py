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


@pytest.mark.parametrize("in_flight", [1, 2, 3])
def test_sequence_driver_same_bytes(in_flight):
    """rt_sequence_*: the same frames through the library's own sequence driver (a replica, stream and thread per frame
    in flight), cameras submitted with the frames; statistics come back in submission order."""
    sd = ra.SceneData.load(os.path.join(G, "bunny.npz"))
    w, h, n = 400, 240, 9
    base = {k: np.array(v, np.float32) for k, v in sd.camera.items()}
    cams = []
    for k in range(n):
        c = dict(base)
        c["pos"] = base["pos"] + np.array([0.2 * k, 0.0, 0.1 * k], np.float32)
        cams.append(c)
    want, want_rays = [], []
    with ra.Scene(sd) as sc:
        for c in cams:
            sc.set_camera(c["pos"], c["up"], c["right"], c["look_at"])
            f, st = sc.render(w, h, 4, abi.RT_COOK_TORRANCE, abi.RT_JITTER_HASH, 2)
            want.append(f.copy()); want_rays.append(st["rays_total"])
    p = ra.make_params(w, h, 4, abi.RT_COOK_TORRANCE, abi.RT_JITTER_HASH, 2)
    bufs = [torch.zeros(h * w * 4, dtype=torch.uint8, device="cuda") for _ in range(n)]
    got_rays = []
    with ra.Sequence(sd, frames_in_flight=in_flight) as seq:
        assert seq.frames_in_flight() == in_flight
        with pytest.raises(ra.RaytracerError):
            seq.wait()                                   # nothing in flight
        for k in range(n):
            if k >= in_flight:
                got_rays.append(seq.wait()["rays_total"])
            seq.submit(p, bufs[k].data_ptr(), cams[k])
        while len(got_rays) < n:
            got_rays.append(seq.wait()["rays_total"])
    torch.cuda.synchronize()
    assert got_rays == want_rays
    for k in range(n):
        assert bufs[k].cpu().numpy().reshape(h, w, 4).tobytes() == want[k].tobytes(), k


def test_sequence_driver_refuses_more_frames_than_it_holds():
    sd = ra.SceneData.load(os.path.join(G, "cornell.npz"))
    p = ra.make_params(64, 48, 1, abi.RT_PHONG, abi.RT_JITTER_HASH, 0)
    bufs = [torch.zeros(64 * 48 * 4, dtype=torch.uint8, device="cuda") for _ in range(3)]
    with ra.Sequence(sd, frames_in_flight=2) as seq:
        seq.submit(p, bufs[0].data_ptr())
        seq.submit(p, bufs[1].data_ptr())
        with pytest.raises(ra.RaytracerError):
            seq.submit(p, bufs[2].data_ptr())
        seq.wait()
        seq.submit(p, bufs[2].data_ptr())
        seq.wait(); seq.wait()
    torch.cuda.synchronize()
    assert bufs[0].cpu().numpy().tobytes() == bufs[1].cpu().numpy().tobytes() == bufs[2].cpu().numpy().tobytes()
    with pytest.raises(ra.RaytracerError):
        ra.Sequence(sd, frames_in_flight=0)


def test_sequence_frame_without_a_camera_takes_the_one_before_it():
    """rt_sequence_submit(camera = NULL) means "the camera of the frame before", whichever replica rendered that one."""
    sd = ra.SceneData.load(os.path.join(G, "cornell.npz"))
    w, h = 160, 120
    base = {k: np.array(v, np.float32) for k, v in sd.camera.items()}
    moved = dict(base, pos=base["pos"] + np.array([1.5, 0.5, 0.0], np.float32))
    with ra.Scene(sd) as sc:
        first, _ = sc.render(w, h, 1, abi.RT_PHONG, abi.RT_JITTER_HASH, 2)
        sc.set_camera(moved["pos"], moved["up"], moved["right"], moved["look_at"])
        second, _ = sc.render(w, h, 1, abi.RT_PHONG, abi.RT_JITTER_HASH, 2)
    assert not np.array_equal(first, second)
    p = ra.make_params(w, h, 1, abi.RT_PHONG, abi.RT_JITTER_HASH, 2)
    bufs = [torch.zeros(h * w * 4, dtype=torch.uint8, device="cuda") for _ in range(4)]
    with ra.Sequence(sd, frames_in_flight=2) as seq:
        got = []
        for k, cam in enumerate((None, moved, None, None)):      # frames 2 and 3 land on different replicas
            seq.submit(p, bufs[k].data_ptr(), cam)
            seq.wait()
            torch.cuda.synchronize()
            got.append(bufs[k].cpu().numpy().reshape(h, w, 4))
    assert np.array_equal(got[0], first)
    for k in (1, 2, 3):
        assert np.array_equal(got[k], second), k
