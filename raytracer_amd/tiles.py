"""Tile partition of a frame across ranks (one process per GPU) and the gather of the parts.

The render path is data-parallel over pixels (kernel.cu:484-543: one thread per sample, no
communication), so N ranks split the frame by 32x32-pixel tiles: tile k (row-major over the tile grid)
belongs to rank k mod N, which renders it with rt_render_params.tile_first = rank, tile_stride = N,
out_layout = RT_OUT_TILES into a compact buffer of rt_tile_count() tiles.  The only exchange is the
gather of those compact buffers to one rank, which un-tiles them with rt_untile_device.

Nothing here computes pixels.  `gather_parts` works on any torch.distributed backend (RCCL on the GPUs,
gloo in the CPU tests); `untile_host` is the host-side statement of rt_untile_device's index map for
callers that gathered to host memory.
"""
import numpy as np

from . import abi

T = abi.RT_TILE


def tile_grid(width, height):
    return (width + T - 1) // T, (height + T - 1) // T


def tiles_of(width, height, rank, world):
    """Tile indices (row-major over the tile grid) that `rank` of `world` renders, in buffer order."""
    tx, ty = tile_grid(width, height)
    return list(range(rank, tx * ty, world))


def part_tiles(width, height, world):
    """Tiles in the largest part: every rank's buffer is padded to this so the gather is regular."""
    tx, ty = tile_grid(width, height)
    return (tx * ty + world - 1) // world


def part_bytes(width, height, world):
    return part_tiles(width, height, world) * T * T * 4


def gather_parts(part, world, rank, dst=0):
    """Gather every rank's compact tile buffer (a 1-D uint8 tensor of part_bytes()) to `dst`.

    Returns the (world * part_bytes) tensor on `dst`, None elsewhere.  One collective per frame; with
    world == 1 there is nothing to exchange and `part` comes back as is."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return part
    parts = None
    gather_list = None
    if rank == dst:
        parts = torch.empty(world * part.numel(), dtype=part.dtype, device=part.device)
        gather_list = list(parts.view(world, part.numel()).unbind(0))
    dist.gather(part, gather_list, dst=dst)
    return parts


class GatherPipeline:
    """Double-buffered gather of the ranks' compact tile buffers: while frame k's parts travel to `dst`, the ranks
    already render frame k + 1 into the other buffer.  `submit(b)` starts the gather of buffer b (asynchronously on
    a backend that can, i.e. RCCL); `collect(b)` makes the current stream wait for it and returns the gathered
    (world * part_bytes) tensor on `dst` (None elsewhere, and None if nothing was submitted for b)."""

    def __init__(self, part_bytes, world, rank, device, dst=0, host_staged=False):
        import torch
        self.world, self.rank, self.dst, self.host_staged = world, rank, dst, host_staged
        self.part = [torch.zeros(part_bytes, dtype=torch.uint8, device=device) for _ in range(2)]
        self.parts = [torch.zeros(world * part_bytes, dtype=torch.uint8, device=device) if rank == dst else None for _ in range(2)]
        self.work = [None, None]
        self.pending = [False, False]

    def submit(self, b):
        import torch.distributed as dist
        if self.host_staged:                      # rehearsal on a backend without device collectives
            got = gather_parts(self.part[b].cpu(), self.world, self.rank, self.dst)
            if self.rank == self.dst:
                self.parts[b].copy_(got)
            self.work[b] = None
        else:
            gl = list(self.parts[b].view(self.world, -1).unbind(0)) if self.rank == self.dst else None
            self.work[b] = dist.gather(self.part[b], gl, dst=self.dst, async_op=True)
        self.pending[b] = True

    def collect(self, b):
        if not self.pending[b]:
            return None
        if self.work[b] is not None:
            self.work[b].wait()
            self.work[b] = None
        self.pending[b] = False
        return self.parts[b]


def untile_host(parts, part_stride_bytes, n_parts, width, height):
    """Row-major (height, width, 4) frame from `n_parts` compact tile buffers laid back to back."""
    parts = np.asarray(parts, dtype=np.uint8).reshape(-1)
    tx, _ = tile_grid(width, height)
    y, x = np.mgrid[0:height, 0:width]
    tile = (y // T) * tx + (x // T)
    part, k = tile % n_parts, tile // n_parts
    px = part.astype(np.int64) * (part_stride_bytes // 4) + k.astype(np.int64) * T * T + (y % T) * T + (x % T)
    return parts.view(np.uint8).reshape(-1, 4)[px.reshape(-1)].reshape(height, width, 4)
