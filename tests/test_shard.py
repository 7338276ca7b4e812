"""Row-tile shards (rt_opts tile_*), host-side scatter, and the world_size-2 gather (gloo)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def test_shard_geometry(rtmi, scenes_dir):
    sc = rtmi.Scene.load(os.path.join(scenes_dir, "three_sphere.json"))
    sc.override(width=32, height=45, spp=1)  # 45 rows: 5 full tiles of 8 + one of 5
    assert sc.shard_rows() == 45
    np.testing.assert_array_equal(sc.shard_global_rows(), np.arange(45))
    seen = []
    for world in (2, 3, 8):
        allrows = []
        for r in range(world):
            rows = sc.shard_global_rows(rtmi.Opts(tile_first=r, tile_stride=world))
            # tile t -> rank t mod world, rows of a tile stay together and ascending
            assert all(((y // 8) % world) == r for y in rows)
            assert list(rows) == sorted(rows)
            allrows += list(rows)
        assert sorted(allrows) == list(range(45))
        seen.append(world)
    # the rotated interleave (rt_opts.tile_rotate: what rt_render_hip_tiles and bench.py use): tile t -> rank (t + t // world)
    # mod world -- one tile of every group of `world` tiles per rank, a different one from group to group
    for world in (2, 3, 4, 8):
        allrows, counts = [], []
        for r in range(world):
            o = rtmi.Opts(tile_first=r, tile_stride=world, tile_rotate=1)
            rows = sc.shard_global_rows(o)
            t = np.asarray(rows) // 8
            assert np.array_equal((t + t // world) % world, np.full(len(rows), r))
            assert list(rows) == sorted(rows) and sc.shard_rows(o) == len(rows)
            if len(set(t)) > 1:
                assert len(set(t % world)) > 1                      # not one row phase for itself
            allrows += list(rows)
            counts.append(len(set(t)))
        assert sorted(allrows) == list(range(45)) and max(counts) - min(counts) <= 1
    # there and back (tile_rotate = 2): groups of 2 * world tiles go to ranks 0 .. world-1, then world-1 .. 0
    for world in (2, 3, 4, 8):
        allrows, counts = [], []
        for r in range(world):
            o = rtmi.Opts(tile_first=r, tile_stride=world, tile_rotate=2)
            rows = sc.shard_global_rows(o)
            t = np.asarray(rows) // 8
            p = t % (2 * world)
            assert np.array_equal(np.where(p < world, p, 2 * world - 1 - p), np.full(len(rows), r))
            assert list(rows) == sorted(rows) and sc.shard_rows(o) == len(rows)
            allrows += list(rows)
            counts.append(len(set(t)))
        assert sorted(allrows) == list(range(45)) and max(counts) - min(counts) <= 1
    with pytest.raises(rtmi.RtmiError):
        sc.shard_rows(rtmi.Opts(tile_first=0, tile_stride=2, tile_rotate=3))
    # which deal rt_render_hip_tiles and bench.py use: rotated once the frame has 4 N^2 tiles, else there and back
    big = rtmi.Scene.rtiow(7, 64, 1080, 1, 5)
    assert [big.shard_deal(None, n) for n in (1, 2, 4, 8)] == [0, 1, 1, 2]
    assert sc.shard_deal(rtmi.Opts(tile_rows=1), 3) == 1 and sc.shard_deal(None, 3) == 2   # 45 rows: 45 tiles of 1 row, 6 of 8
    # tile_rows other than 8, partial last tile
    rows = sc.shard_global_rows(rtmi.Opts(tile_rows=16, tile_first=1, tile_stride=2))
    assert list(rows) == list(range(16, 32))
    rows = sc.shard_global_rows(rtmi.Opts(tile_rows=16, tile_first=0, tile_stride=2))
    assert list(rows) == list(range(0, 16)) + list(range(32, 45))
    # a rank beyond the last tile owns nothing
    assert sc.shard_rows(rtmi.Opts(tile_rows=32, tile_first=5, tile_stride=8)) == 0
    with pytest.raises(rtmi.RtmiError):
        sc.shard_rows(rtmi.Opts(tile_first=3, tile_stride=2))


def test_host_scatter_rows(rtmi, scenes_dir):
    sc = rtmi.Scene.load(os.path.join(scenes_dir, "three_sphere.json"))
    sc.override(width=16, height=21, spp=1)
    full = np.arange(21 * 16 * 3, dtype=np.float32).reshape(21, 16, 3)
    out = np.zeros_like(full)
    for r in range(3):
        o = rtmi.Opts(tile_rows=4, tile_first=r, tile_stride=3)
        rows = sc.shard_global_rows(o)
        sc.scatter_rows(o, full[rows], out)
    np.testing.assert_array_equal(out, full)


def _worker(rank, world, port, height, tile_rows, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    from __graft_entry__ import load_package
    rtmi = load_package()
    import importlib
    rdist = importlib.import_module("rtmi.dist")
    import rtcheck
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc = rtmi.Scene.load(os.path.join(root, "ray-tracing-in-cuda_amd", "scenes", "three_sphere.json"))
        sc.override(width=24, height=height, spp=2)
        base = rtmi.Opts(seed=11, tile_rows=tile_rows)
        mine = rdist.shard_opts(base, rank, world)
        # stand-in for the GPU render of this rank's rows (CPU test): the checker's image rows
        full_ref, _ = rtcheck.oracle_render(sc, seed=11, threads=1)
        local = rdist.alloc_local(sc, base, world, "cpu")
        rows = sc.shard_global_rows(mine)
        local[: len(rows)] = torch.from_numpy(full_ref[rows])
        full = rdist.gather_framebuffer(local, sc, base, rank, world)
        ok = True
        if rank == 0:
            ok = bool(np.array_equal(full.numpy(), full_ref))
        else:
            assert full is None
        # the two-phase form bench.py pipelines (a frame's gather travels while the next frame renders): two frames in
        # flight in two local buffers and two receive slots, placed in order; every deal of the tiles (rt_opts.tile_rotate)
        for deal in (0, 1, 2):
            b2 = rdist.shard_opts(base, 0, 1)
            b2.tile_rotate = deal
            m2 = rdist.shard_opts(b2, rank, world)
            rows2 = sc.shard_global_rows(m2)
            bufs = [rdist.alloc_local(sc, b2, world, "cpu") for _ in range(2)]
            frames = [full_ref, full_ref * 2.0 + 1.0]
            pend = []
            for i in range(2):
                bufs[i][: len(rows2)] = torch.from_numpy(frames[i][rows2])
                pend.append(rdist.gather_begin(bufs[i], rank, world, slot=i))
            for i in range(2):
                got = rdist.gather_end(pend[i], sc, b2, rank, world, bufs[i].shape[0], "cpu")
                if rank == 0:
                    ok = ok and bool(np.array_equal(got.numpy(), frames[i]))
                else:
                    assert got is None
        if rank == 0:
            q.put(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("height,tile_rows", [(32, 8), (29, 8), (20, 4)])
def test_gather_world2_gloo(height, tile_rows):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + height + tile_rows) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, height, tile_rows, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_gather_world1_no_comm(rtmi, scenes_dir):
    import importlib
    rdist = importlib.import_module("rtmi.dist")
    sc = rtmi.Scene.load(os.path.join(scenes_dir, "three_sphere.json"))
    sc.override(width=8, height=10, spp=1)
    base = rtmi.Opts()
    local = torch.arange(10 * 8 * 3, dtype=torch.float32).reshape(10, 8, 3)
    full = rdist.gather_framebuffer(local, sc, base, 0, 1)
    assert torch.equal(full, local)
