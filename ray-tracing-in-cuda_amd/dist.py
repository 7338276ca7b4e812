"""Row-tile sharding of one frame over the GPUs of a node + the framebuffer gather.

The reference's only multi-GPU mechanism is one renderer PROCESS per GPU per animation
frame (gpu-version/blue.py:23-32, CUDA_VISIBLE_DEVICES=k); nothing is exchanged.  Here one
frame is split: row tile t (tile_rows full-width rows) belongs to rank t mod world, so
sky rows and ground rows are spread evenly; pixels are independent and the RNG is keyed
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


def shard_opts(base: Opts, rank: int, world: int) -> Opts:
    """rt_opts of one rank: interleaved row tiles rank, rank+world, ..."""
    o = Opts()
    for name, _ in Opts._fields_:
        setattr(o, name, getattr(base, name))
    o.tile_first = rank
    o.tile_stride = world
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
    parts = None
    send = local.cpu() if via_host else local
    if rank == dst:
        parts = [torch.empty_like(send) for _ in range(world)]
    dist.gather(send, gather_list=parts, dst=dst, group=group)
    if rank != dst:
        return None
    if via_host:
        parts = [p.to(local.device, non_blocking=True) for p in parts]
    full = out if out is not None else torch.empty((scene.height, scene.width, 3), dtype=local.dtype,
                                                   device=local.device)
    for r in range(world):
        rows = _rows_on(scene, base, r, world, local.device)
        if len(rows):
            full.index_copy_(0, rows, parts[r][: len(rows)])
    return full


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
