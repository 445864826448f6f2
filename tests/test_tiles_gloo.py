"""The N > 1 plan on CPU: two gloo ranks, the frame split by tiles, one gather, un-tile on rank 0.

No GPU here, so each rank takes its tiles out of a frame the ORACLE rendered (the oracle stands in for
the per-rank render, as the checker's picture of the right answer); what is under test is the partition,
the padding of the parts, the gather and the un-tile index map that bench.py uses with RCCL.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H = 100, 70


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _pack(frame, rank, world):
    from raytracer_amd import tiles
    buf = np.zeros(tiles.part_bytes(W, H, world), np.uint8)
    tx, _ = tiles.tile_grid(W, H)
    for k, tile in enumerate(tiles.tiles_of(W, H, rank, world)):
        x0, y0 = (tile % tx) * 32, (tile // tx) * 32
        blk = np.zeros((32, 32, 4), np.uint8)
        sub = frame[y0:y0 + 32, x0:x0 + 32]
        blk[:sub.shape[0], :sub.shape[1]] = sub
        buf[k * 4096:(k + 1) * 4096] = blk.reshape(-1)
    return buf


def _worker(rank, world, port, out_path):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    import orclib
    import raytracer_amd as ra
    from raytracer_amd import abi, tiles
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sd = ra.SceneData.load(os.path.join(ROOT, "tests", "golden", "scenes", "cornell.npz"))
        osc = orclib.OracleScene(sd)
        frame = osc.render(W, H, 1, abi.RT_PHONG, abi.RT_JITTER_HASH, 0, 0, 1, want_samples=False, want_primary=False)["rgba"]
        osc.close()
        part = torch.from_numpy(_pack(frame, rank, world))
        parts = tiles.gather_parts(part, world, rank, dst=0)
        # whole-job ray count the way bench.py forms it: a sum over ranks
        mine = torch.tensor([float(len(tiles.tiles_of(W, H, rank, world)))], dtype=torch.float64)
        dist.all_reduce(mine, op=dist.ReduceOp.SUM)
        # the double-buffered form bench.py uses: three frames in flight over two buffers, collected in order
        pipe = tiles.GatherPipeline(tiles.part_bytes(W, H, world), world, rank, "cpu", dst=0)
        piped_ok = True
        frames = [frame, frame[::-1].copy(), np.roll(frame, 3, axis=1)]
        for k, f in enumerate(frames + [None, None]):
            b = k % 2
            got = pipe.collect(b)
            if got is not None and rank == 0:
                want = frames[k - 2]
                piped_ok &= bool((tiles.untile_host(got.numpy(), tiles.part_bytes(W, H, world), world, W, H) == want).all())
            if f is not None:
                pipe.part[b].copy_(torch.from_numpy(_pack(f, rank, world)))
                pipe.submit(b)
        if rank == 0:
            got = tiles.untile_host(parts.numpy(), tiles.part_bytes(W, H, world), world, W, H)
            np.savez(out_path, ok=np.array(bool((got == frame).all()) and piped_ok), tiles=np.array(mine.item()),
                     nonzero=np.array(int((frame[:, :, :3] != 0).sum())))
        else:
            assert parts is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_tile_split_gather_untile(tmp_path, world):
    import torch.multiprocessing as mp
    from raytracer_amd import tiles
    out = str(tmp_path / "r0.npz")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    z = np.load(out)
    assert bool(z["ok"]) and int(z["nonzero"]) > 0
    tx, ty = tiles.tile_grid(W, H)
    assert int(z["tiles"]) == tx * ty
