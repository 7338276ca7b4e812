"""-m gpu: every BASELINE.json configuration at its FULL frame geometry (tile / band counts, accumulator
plane size, chunk schedule), through the C ABI, against the CPU checker (oracle/rt_oracle.c).

A full frame at full spp is too much for the CPU checker, so each configuration is covered by
  * the whole frame at 1-2 spp: determinism, 8-shard row-tile assembly == unsharded, culled kernel ==
    linear scan (variant 16), and bit-equality with the checker on sampled rows (bottom, top, interior);
  * the LAST sample index of the configuration (sample_first = spp - 1) on the whole frame;
  * the configuration's full-spp chunk schedule (big / medium / small runs of work items) on ONE row tile,
    checked against two accumulated halves (exact sums) and against the checker on a pixel window.
Configs: BASELINE.json configs[0..4] (C1, C2a/C2b, C3, C4, C5).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED = 2023


def _rows_equal_checker(rtcheck, sc, full, bands, seed=SEED, **kw):
    osc = rtcheck.OracleScene(sc)
    for y0, y1 in bands:
        ref, _ = rtcheck.oracle_render(osc, seed=seed, rows=(y0, y1), **kw)
        assert np.array_equal(full[y0:y1], ref[y0:y1]), f"rows {y0}..{y1} differ from the CPU checker"


def _assemble(rtmi, sc, world, seed=SEED, **kw):
    out = np.zeros((sc.height, sc.width, 3), dtype=np.float32)
    rows = 0
    for r in range(world):
        o = rtmi.Opts(seed=seed, tile_first=r, tile_stride=world, **kw)
        local = sc.render(o)
        rows += local.shape[0]
        sc.scatter_rows(o, local, out)
    assert rows == sc.height
    return out


def _full_schedule_on_one_tile(rtmi, rtcheck, sc, tile, window, seed=SEED):
    """The configuration's own spp (its chunk schedule: runs of 64-, 16- and 4-sample work items) on row tile
    `tile` alone; `window` = (x0, x1, ly0, ly1) in tile-local rows is checked against the CPU checker."""
    spp = sc.spp
    one = rtmi.Opts(seed=seed, tile_first=tile, tile_stride=1 << 20)
    assert sc.shard_rows(one) == min(8, sc.height - 8 * tile)
    st = rtmi.Stats()
    img = sc.render(one, st)
    assert st.kernel_ms > 0
    # any split of the sample range adds up to the same exact sums
    half = spp // 2
    a = rtmi.Opts(seed=seed, tile_first=tile, tile_stride=1 << 20, sample_first=0, sample_count=half)
    b = rtmi.Opts(seed=seed, tile_first=tile, tile_stride=1 << 20, sample_first=half, sample_count=spp - half)
    acc, _ = sc.accumulate(None, a, want_image=False)
    acc, two = sc.accumulate(acc, b)
    assert np.array_equal(two, img)
    # a different work-item size is a different schedule of the same sums
    assert np.array_equal(img, sc.render(rtmi.Opts(seed=seed, tile_first=tile, tile_stride=1 << 20, spp_chunk=16)))
    x0, x1, ly0, ly1 = window
    y0 = 8 * tile
    ref = rtcheck.oracle_render_rect(sc, seed, x0, x1, y0 + ly0, y0 + ly1, sample_count=spp)
    assert np.array_equal(img[ly0:ly1, x0:x1], ref), "full-spp window differs from the CPU checker"
    return img


def test_config5_rtiow_4k(rtmi, rtcheck):
    """configs[4]: random-spheres 3840x2160, 8192 spp, depth 50, row tiles over 8 GPUs.
    480 x 270 wave tiles, 199 MB accumulator plane, 128 + tail sample chunks."""
    sc = rtmi.Scene.rtiow(7, 3840, 2160, 1, 50)
    full = sc.render(rtmi.Opts(seed=SEED))
    assert full.shape == (2160, 3840, 3)
    assert np.array_equal(full, sc.render(rtmi.Opts(seed=SEED)))                 # deterministic
    assert np.array_equal(full, sc.render(rtmi.Opts(seed=SEED, variant=16)))     # culled == linear scan
    assert np.array_equal(full, _assemble(rtmi, sc, 8))                          # 8 row-tile shards == unsharded
    assert sc.shard_rows(rtmi.Opts(tile_first=7, tile_stride=8)) == 264          # 33 tiles of 8 rows
    bands = [(0, 2), (2158, 2160), (700, 702), (1403, 1405)]
    _rows_equal_checker(rtcheck, sc, full, bands)
    mean = full.astype(np.float64).mean()
    assert 0.3 < mean < 0.7 and np.isfinite(full).all() and full.min() >= 0
    # the last sample index of the 8192-spp run, whole frame
    last = sc.render(rtmi.Opts(seed=SEED, sample_first=8191, sample_count=1))
    assert not np.array_equal(last, full)
    _rows_equal_checker(rtcheck, sc, last, [(0, 1), (1079, 1080), (2159, 2160)], sample_first=8191, sample_count=1)
    # the 8192-spp schedule (125 + tail items per tile) on one interior row tile
    sc.override(spp=8192)
    _full_schedule_on_one_tile(rtmi, rtcheck, sc, tile=60, window=(1917, 1921, 3, 5))


def test_config4_sample_scene_1080p(rtmi, rtcheck, golden_dir):
    """configs[3]: gpu-version/sample_scene.json (5 spheres + a rotated cylinder, constant background) at
    1920x1080, 4096 spp."""
    sc = rtmi.Scene.load(os.path.join(golden_dir, "scenes", "sample_scene.json"))
    sc.override(width=1920, height=1080, spp=2, max_depth=50)
    full = sc.render(rtmi.Opts(seed=SEED))
    assert np.array_equal(full, sc.render(rtmi.Opts(seed=SEED)))
    assert np.array_equal(full, sc.render(rtmi.Opts(seed=SEED, variant=16)))
    assert np.array_equal(full, _assemble(rtmi, sc, 8))
    assert np.array_equal(full, _assemble(rtmi, sc, 3, tile_rows=16))
    _rows_equal_checker(rtcheck, sc, full, [(0, 2), (1078, 1080), (400, 402), (641, 643)])
    last = sc.render(rtmi.Opts(seed=SEED, sample_first=4095, sample_count=1))
    _rows_equal_checker(rtcheck, sc, last, [(539, 541)], sample_first=4095, sample_count=1)
    sc.override(spp=4096)
    _full_schedule_on_one_tile(rtmi, rtcheck, sc, tile=67, window=(956, 964, 2, 4))


def test_config3_rtiow_1080p_full_schedule(rtmi, rtcheck):
    """configs[2] (the bench workload): the 1024-spp schedule on one row tile; the whole frame at 2 spp is
    test_gpu_parity.py::test_full_frame_properties."""
    sc = rtmi.Scene.rtiow(7, 1920, 1080, 1024, 50)
    _full_schedule_on_one_tile(rtmi, rtcheck, sc, tile=40, window=(700, 708, 0, 2))
    last = rtmi.Scene.rtiow(7, 1920, 1080, 1, 50).render(rtmi.Opts(seed=SEED, sample_first=1023, sample_count=1))
    _rows_equal_checker(rtcheck, rtmi.Scene.rtiow(7, 1920, 1080, 1, 50), last, [(300, 302)], sample_first=1023,
                        sample_count=1)


def test_config2_basic_scene_and_dna_720p(rtmi, rtcheck, golden_dir):
    """configs[1]: gpu-version/basic_scene.json at 1280x720, 256 spp.  As shipped its object list is empty
    (C2a: pure background); dna.py fills it per frame (C2b: 60 emissive spheres + 30 emissive cylinders)."""
    empty = rtmi.Scene.load(os.path.join(golden_dir, "scenes", "basic_scene.json"))
    empty.override(width=1280, height=720, spp=256, max_depth=50)
    img = empty.render(rtmi.Opts(seed=SEED))
    bg = np.array(list(empty.info.background), dtype=np.float32)
    # every sample is the constant background: 256 x its 2^-24 fixed-point value, converted to fp32 once
    want = (np.rint(bg.astype(np.float64) * 2.0 ** 24) * 256 * 2.0 ** -24).astype(np.float32)
    assert np.array_equal(img, np.broadcast_to(want, img.shape))
    _rows_equal_checker(rtcheck, empty, img, [(0, 1), (719, 720)])

    dna = rtmi.Scene.dna(0.0)  # frame 0 of dna.py
    dna.override(width=1280, height=720, spp=4, max_depth=50)
    full = dna.render(rtmi.Opts(seed=SEED))
    assert np.array_equal(full, dna.render(rtmi.Opts(seed=SEED)))
    assert np.array_equal(full, dna.render(rtmi.Opts(seed=SEED, variant=16)))
    assert np.array_equal(full, _assemble(rtmi, dna, 8))
    _rows_equal_checker(rtcheck, dna, full, [(0, 2), (718, 720), (359, 361), (200, 202)])
    assert full.max() > 0
    dna.override(spp=256)
    _full_schedule_on_one_tile(rtmi, rtcheck, dna, tile=45, window=(600, 680, 0, 8))


def test_config1_three_sphere_full(rtmi, rtcheck, scenes_dir):
    """configs[0]: the cmake-cpu-version 3-sphere scene, 400x225, 100 spp, depth 50 -- the whole
    configuration, bit for bit."""
    sc = rtmi.Scene.load(os.path.join(scenes_dir, "three_sphere.json"))
    sc.override(width=400, height=225, spp=100, max_depth=50)
    img = sc.render(rtmi.Opts(seed=SEED))
    ref, _ = rtcheck.oracle_render(sc, seed=SEED)
    assert np.abs(img - ref).max() / 100 < 1e-3  # north_star tolerance
    assert np.array_equal(img, ref)
    assert np.array_equal(img, _assemble(rtmi, sc, 8))
