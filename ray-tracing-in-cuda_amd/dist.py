"""Row-tile sharding of one frame over the GPUs of a node + the framebuffer gather.

The reference's only multi-GPU mechanism is one renderer PROCESS per GPU per animation
frame (gpu-version/blue.py:23-32, CUDA_VISIBLE_DEVICES=k); nothing is exchanged.  Here one
frame is split: row tile t (tile_rows full-width rows) belongs to rank (t + t // world) mod world
(an interleave whose phase rotates from one group of `world` tiles to the next), so sky rows and
ground rows are spread evenly and no rank keeps one row phase of the image; pixels are independent and the RNG is keyed
by the GLOBAL pixel id, so the assembled image is bit-identical for every world size.
The only communication is ONE gather of the rank-local row buffers to the root
(torch.distributed.gather -> ncclGather over xGMI with the nccl (= RCCL) backend, gloo in
the CPU tests); the root then scatters rows to their image positions.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from . import Opts


def shard_opts(base: Opts, rank: int, world: int, deal: int | None = None) -> Opts:
    """rt_opts of one rank: its share of the frame's row tiles.  `deal` = rt_opts.tile_rotate (include/rtmi.h: 0 plain
    interleave, 1 rotated, 2 there and back); None keeps base.tile_rotate when it is set, else the rotated interleave --
    callers that know the scene pass scene.shard_deal(base, world), what rt_render_hip_tiles uses."""
    o = Opts()
    for name, _ in Opts._fields_:
        setattr(o, name, getattr(base, name))
    o.tile_first = rank
    o.tile_stride = world
    if world <= 1:
        o.tile_rotate = 0
    elif deal is not None:
        o.tile_rotate = deal
    elif not base.tile_rotate:
        o.tile_rotate = 1
    return o


def max_shard_rows(scene, base: Opts, world: int) -> int:
    return max(scene.shard_rows(shard_opts(base, r, world)) for r in range(world))


def alloc_local(scene, base: Opts, world: int, device) -> torch.Tensor:
    """Rank-local framebuffer, padded to the largest shard so the gather is uniform."""
    return torch.zeros((max_shard_rows(scene, base, world), scene.width, 3), dtype=torch.float32, device=device)


def gather_framebuffer(local: torch.Tensor, scene, base: Opts, rank: int, world: int, dst: int = 0, group=None,
                       out: torch.Tensor | None = None, via_host: bool = False):
    """One gather of every rank's (padded) local rows to `dst`; returns the assembled
    (H, W, 3) image on dst, None elsewhere.  world == 1 needs no communication.
    via_host: stage through host memory (gloo rehearsals of the multi-rank path)."""
    if world == 1:
        full = out if out is not None else torch.empty((scene.height, scene.width, 3), dtype=local.dtype,
                                                       device=local.device)
        rows = _rows_on(scene, base, 0, 1, local.device)
        full.index_copy_(0, rows, local[: len(rows)])
        return full
    send = local.cpu() if via_host else local
    big = None
    parts = None
    if rank == dst:
        # one receive buffer for all ranks (cached: the gather runs every step), gathered into as views
        big = _recv_buffer(send, world)
        parts = list(big.unbind(0))
    dist.gather(send, gather_list=parts, dst=dst, group=group)
    if rank != dst:
        return None
    if via_host:
        big = big.to(local.device, non_blocking=True)
    full = out if out is not None else torch.empty((scene.height, scene.width, 3), dtype=local.dtype,
                                                   device=local.device)
    # one scatter for all shards: source rows (rank r's valid rows inside the padded receive buffer) -> image rows
    src, dst_rows = _gather_indices(scene, base, world, local.shape[0], local.device)
    full.index_copy_(0, dst_rows, big.view(-1, scene.width, 3).index_select(0, src))
    return full


class PendingGather:
    """A gather that has been started (gather_begin) and not yet placed (gather_end)."""
    __slots__ = ("work", "big", "send")

    def __init__(self, work, big, send):
        self.work, self.big, self.send = work, big, send


def gather_begin(local: torch.Tensor, rank: int, world: int, dst: int = 0, group=None, via_host: bool = False,
                 slot: int = 0) -> PendingGather:
    """First half of gather_framebuffer for world > 1: start the gather of `local` and return without making the current
    stream wait for it, so that the caller's next launch (the next frame, rendered into ANOTHER local buffer) overlaps the
    transfer.  `slot` selects the root's receive buffer: frames in flight at the same time need different slots."""
    send = local.cpu() if via_host else local
    big = _recv_buffer(send, world, slot) if rank == dst else None
    parts = list(big.unbind(0)) if big is not None else None
    work = dist.gather(send, gather_list=parts, dst=dst, group=group, async_op=True)
    return PendingGather(work, big, send)


def gather_end(p: PendingGather, scene, base: Opts, rank: int, world: int, padded_rows: int, device, dst: int = 0,
               out: torch.Tensor | None = None, via_host: bool = False):
    """Second half: wait for the gather (the current stream waits, not the host, for nccl) and place the rows on dst.
    After it the local buffer the gather read may be written again."""
    p.work.wait()
    if rank != dst:
        return None
    big = p.big.to(device, non_blocking=True) if via_host else p.big
    full = out if out is not None else torch.empty((scene.height, scene.width, 3), dtype=big.dtype, device=device)
    src, dst_rows = _gather_indices(scene, base, world, padded_rows, device)
    full.index_copy_(0, dst_rows, big.view(-1, scene.width, 3).index_select(0, src))
    return full


_RECV_CACHE: dict = {}


def _recv_buffer(like: torch.Tensor, world: int, slot: int = 0) -> torch.Tensor:
    key = (tuple(like.shape), like.dtype, str(like.device), world, slot)
    t = _RECV_CACHE.get(key)
    if t is None:
        t = torch.empty((world,) + tuple(like.shape), dtype=like.dtype, device=like.device)
        _RECV_CACHE[key] = t
    return t


_INDEX_CACHE: dict = {}


def _gather_indices(scene, base: Opts, world: int, padded_rows: int, device):
    key = (id(scene), scene.height, base.tile_rows, world, padded_rows, str(device))
    t = _INDEX_CACHE.get(key)
    if t is None:
        src, dst_rows = [], []
        for r in range(world):
            rows = scene.shard_global_rows(shard_opts(base, r, world))
            src.append(np.arange(len(rows), dtype=np.int64) + r * padded_rows)
            dst_rows.append(np.asarray(rows, dtype=np.int64))
        t = (torch.as_tensor(np.concatenate(src), device=device), torch.as_tensor(np.concatenate(dst_rows), device=device))
        _INDEX_CACHE[key] = t
    return t


_ROWS_CACHE: dict = {}


def _rows_on(scene, base: Opts, r: int, world: int, device) -> torch.Tensor:
    """Global row indices of rank r's shard as a tensor on `device` (cached: the gather runs every step)."""
    key = (id(scene), scene.height, base.tile_rows, r, world, str(device))
    t = _ROWS_CACHE.get(key)
    if t is None:
        t = torch.as_tensor(scene.shard_global_rows(shard_opts(base, r, world)), device=device)
        _ROWS_CACHE[key] = t
    return t


def shard_row_table(scene, base: Opts, world: int) -> list[np.ndarray]:
    return [scene.shard_global_rows(shard_opts(base, r, world)) for r in range(world)]
