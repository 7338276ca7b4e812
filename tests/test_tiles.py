"""rt_render_hip_tiles: one frame on several GPUs behind the C ABI (row tiles interleaved over the devices, ONE
ncclGather to the first, a placement kernel there).  A test box has one GPU: the n = 1 call runs the whole path
-- per-device stream, ncclCommInitAll, ncclGather, placement kernel -- and the N > 1 row placement is checked by
laying N shards out the way the gather delivers them."""
import numpy as np
import pytest

SEED = 2023


def test_tiles_needs_a_device(rtmi):
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a HIP device is present: the error path is for boxes without one")
    sc = rtmi.Scene.rtiow(7, 32, 18, 1, 5)
    with pytest.raises(rtmi.RtmiError) as e:
        sc.render_tiles(n=1)
    assert e.value.status == 5  # RT_ERR_HIP: no CPU fallback


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,spp,tile_rows", [(96, 54, 4, 8), (130, 45, 3, 4), (64, 7, 2, 16), (400, 225, 8, 8)])
def test_tiles_one_device_equals_rt_render_hip(rtmi, w, h, spp, tile_rows):
    sc = rtmi.Scene.rtiow(7, w, h, spp, 50)
    want = sc.render(rtmi.Opts(seed=SEED))
    st = rtmi.Stats()
    got = sc.render_tiles([0], rtmi.Opts(seed=SEED, tile_rows=tile_rows), st)
    assert np.array_equal(got, want)
    assert st.devices_used == 1 and st.local_rows == h and st.kernel_ms > 0 and st.gather_ms >= 0
    # devices = NULL means ordinals 0..n-1; buffers and the communicator are reused by the next call
    assert np.array_equal(sc.render_tiles(None, rtmi.Opts(seed=SEED, tile_rows=tile_rows), n=1), want)


@pytest.mark.gpu
def test_tiles_argument_errors(rtmi):
    sc = rtmi.Scene.rtiow(7, 32, 18, 1, 5)
    nd = rtmi.device_count()
    with pytest.raises(rtmi.RtmiError, match="requested"):
        sc.render_tiles(n=nd + 1)
    with pytest.raises(rtmi.RtmiError, match="requested"):
        sc.render_tiles(n=0)
    with pytest.raises(rtmi.RtmiError, match="out of range"):
        sc.render_tiles([nd + 3])
    if nd >= 2:
        with pytest.raises(rtmi.RtmiError, match="twice"):
            sc.render_tiles([0, 0])


def _owner(t, world, rotate):
    """rank of row tile t (include/rtmi.h, rt_opts.tile_rotate)"""
    if rotate == 1:
        return (t + t // world) % world
    if rotate == 2:
        p = t % (2 * world)
        return np.where(p < world, p, 2 * world - 1 - p)
    return t % world


@pytest.mark.gpu
@pytest.mark.parametrize("rotate", [2, 1, 0])
@pytest.mark.parametrize("world,tile_rows,h", [(2, 8, 54), (3, 8, 45), (8, 8, 1080 // 4), (5, 16, 77), (8, 4, 30)])
def test_gathered_layout_placement_kernel(rtmi, world, tile_rows, h, rotate):
    """What the root does after the gather, for N > 1: shards laid out [rank][pad_rows][W][3] -> the frame.  rotate = 1 is
    the split rt_render_hip_tiles and bench.py use (rt_opts.tile_rotate: tile t belongs to shard (t + t // N) mod N)."""
    import torch
    sc = rtmi.Scene.rtiow(7, 120, h, 2, 50)
    want = sc.render(rtmi.Opts(seed=SEED))
    shards = [rtmi.Opts(seed=SEED, tile_rows=tile_rows, tile_first=r, tile_stride=world, tile_rotate=rotate) for r in range(world)]
    owners = np.full(h, -1)
    for r, o in enumerate(shards):  # every row belongs to exactly one shard; tile t to shard (t + t // N) % N or t % N
        rows = sc.shard_global_rows(o)
        assert (owners[rows] == -1).all()
        owners[rows] = r
        t = np.asarray(rows) // tile_rows
        assert np.array_equal(_owner(t, world, rotate), np.full(len(rows), r))
    assert (owners >= 0).all()
    pad = max(sc.shard_rows(o) for o in shards)
    gathered = torch.full((world, pad, 120, 3), float("nan"), dtype=torch.float32, device="cuda:0")
    for r, o in enumerate(shards):
        rows = sc.shard_rows(o)
        if rows:
            sc.render_device(o, gathered[r].data_ptr(), torch.cuda.current_stream().cuda_stream)
    full = torch.empty((h, 120, 3), dtype=torch.float32, device="cuda:0")
    sc.place_rows_device(rtmi.Opts(tile_rows=tile_rows, tile_rotate=rotate), world, pad, gathered.data_ptr(), full.data_ptr(),
                         torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(full.cpu().numpy(), want)
    with pytest.raises(rtmi.RtmiError, match="pad_rows"):
        sc.place_rows_device(rtmi.Opts(tile_rows=tile_rows, tile_rotate=rotate), world, pad - 1, gathered.data_ptr(), full.data_ptr(), 0)


@pytest.mark.gpu
def test_cli_gpus_flag_writes_the_same_ppm(tmp_path):
    """`rtmi --gpus 1` goes through rt_render_hip_tiles (RCCL loaded by the C++ program itself, no PyTorch in the
    process) and writes the bytes of the plain single-device run."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "ray-tracing-in-cuda_amd", "rtmi")
    a, b = tmp_path / "a.ppm", tmp_path / "b.ppm"
    common = ["--rtiow", "-w", "100", "-h", "57", "-spp", "6", "--no-png"]
    r = subprocess.run([exe] + common + ["-o", str(a)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe] + common + ["-o", str(b), "--gpus", "1", "--tile-rows", "4"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "tiles: 1 device(s)" in r.stderr, r.stderr
    assert a.read_bytes() == b.read_bytes()
    r = subprocess.run([exe] + common + ["-o", str(b), "--gpus", "64"], capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "requested" in r.stderr
