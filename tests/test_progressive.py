"""Progressive / resumable rendering: exact int64 pixel sums carried across calls and runs.

CPU part: the conversion helper and the CLI's accumulator-file checks.  GPU part: any split of a sample
range gives the bits of one render over the whole range (through the C ABI and through the `rtmi` binary)."""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RTMI_BIN = os.path.join(ROOT, "ray-tracing-in-cuda_amd", "rtmi")


def test_acc_to_rgb_is_one_rounding_of_the_exact_sum(rtmi):
    acc = np.array([0, 1 << 24, -(1 << 24), 3 << 23, (1 << 24) + 1, 12345678901234567, -7], dtype=np.int64)
    got = rtmi.acc_to_rgb(acc)
    want = (acc.astype(np.float64) / 16777216.0).astype(np.float32)  # exact in double below 2^53
    assert got.dtype == np.float32 and np.array_equal(got, want)
    assert got[1] == 1.0 and got[2] == -1.0 and got[3] == 1.5


def test_accumulate_rejects_bad_buffers(rtmi):
    sc = rtmi.Scene.rtiow(7, 16, 8, 2, 5)
    with pytest.raises(ValueError):
        sc.accumulate(np.zeros((8, 16, 3), dtype=np.float32))
    with pytest.raises(ValueError):
        sc.accumulate(np.zeros((4, 16, 3), dtype=np.int64))


def _run(args, **kw):
    return subprocess.run([RTMI_BIN] + args, capture_output=True, text=True, timeout=300, **kw)


def test_cli_checks_the_accumulator_file_before_rendering(tmp_path):
    bad = tmp_path / "bad.bin"
    bad.write_bytes(b"not an accumulator")
    r = _run(["--rtiow", "-w", "16", "-h", "8", "-spp", "2", "--acc-in", str(bad), "-o", str(tmp_path / "o.ppm")])
    assert r.returncode == 1 and "not an accumulator file" in r.stderr
    # a file of the earlier format (sums in units of 2^-32): refused, not continued 256 x too bright
    old = tmp_path / "old.bin"
    old.write_bytes(struct.pack("<8siiqQq", b"RTMIACC1", 16, 8, 4, 2023, 0) + bytes(16 * 8 * 3 * 8))
    r = _run(["--rtiow", "-w", "16", "-h", "8", "-spp", "2", "--acc-in", str(old), "-o", str(tmp_path / "o.ppm")])
    assert r.returncode == 1 and "fixed-point scale" in r.stderr
    other = tmp_path / "other.bin"
    other.write_bytes(struct.pack("<8siiqQq", b"RTMIACC2", 16, 8, 4, 2023, 32) + bytes(16 * 8 * 3 * 8))
    r = _run(["--rtiow", "-w", "16", "-h", "8", "-spp", "2", "--acc-in", str(other), "-o", str(tmp_path / "o.ppm")])
    assert r.returncode == 1 and "fixed-point scale" in r.stderr
    # right magic, wrong frame size
    hdr = struct.pack("<8siiqQq", b"RTMIACC2", 32, 8, 4, 2023, 24)
    wrong = tmp_path / "wrong.bin"
    wrong.write_bytes(hdr + bytes(32 * 8 * 3 * 8))
    r = _run(["--rtiow", "-w", "16", "-h", "8", "-spp", "2", "--acc-in", str(wrong), "-o", str(tmp_path / "o.ppm")])
    assert r.returncode == 1 and "32x8" in r.stderr
    # matching header, truncated payload
    hdr = struct.pack("<8siiqQq", b"RTMIACC2", 16, 8, 4, 2023, 24)
    short = tmp_path / "short.bin"
    short.write_bytes(hdr + bytes(100))
    r = _run(["--rtiow", "-w", "16", "-h", "8", "-spp", "2", "--acc-in", str(short), "-o", str(tmp_path / "o.ppm")])
    assert r.returncode == 1 and "truncated" in r.stderr
    # sums hold [0,4) but the caller asks to continue at 6
    ok = tmp_path / "ok.bin"
    ok.write_bytes(hdr + bytes(16 * 8 * 3 * 8))
    r = _run(["--rtiow", "-w", "16", "-h", "8", "-spp", "2", "--acc-in", str(ok), "--spp-begin", "6",
              "-o", str(tmp_path / "o.ppm")])
    assert r.returncode == 1 and "--spp-begin" in r.stderr
    assert not (tmp_path / "o.ppm").exists()


@pytest.mark.gpu
def test_any_split_of_the_samples_equals_one_render(rtmi, rtcheck):
    sc = rtmi.Scene.rtiow(7, 96, 56, 24, 50)
    whole = sc.render(rtmi.Opts(seed=11))
    acc = None
    img = None
    for first, n in [(0, 5), (5, 1), (6, 16), (22, 2)]:
        acc, img = sc.accumulate(acc, rtmi.Opts(seed=11, sample_first=first, sample_count=n))
    assert np.array_equal(img, whole)
    assert np.array_equal(rtmi.acc_to_rgb(acc), whole)
    # order of the pieces does not matter either (integer sums commute)
    acc2 = None
    for first, n in [(6, 16), (22, 2), (0, 5), (5, 1)]:
        acc2, _ = sc.accumulate(acc2, rtmi.Opts(seed=11, sample_first=first, sample_count=n), want_image=False)
    assert np.array_equal(acc2, acc)
    # the sums are what the CPU checker accumulates
    osc = rtcheck.OracleScene(sc)
    ref, _ = rtcheck.oracle_render(osc, seed=11)
    assert np.array_equal(img, ref)


@pytest.mark.gpu
def test_accumulate_on_a_row_shard_and_with_emitters(rtmi):
    sc = rtmi.Scene.load(os.path.join(ROOT, "ray-tracing-in-cuda_amd", "scenes", "mixed_emissive.json"))
    sc.override(64, 48, 12, 8)
    o = rtmi.Opts(seed=5, tile_rows=8, tile_first=1, tile_stride=2)
    whole = sc.render(o)
    acc, _ = sc.accumulate(None, rtmi.Opts(seed=5, tile_rows=8, tile_first=1, tile_stride=2, sample_first=0, sample_count=7),
                           want_image=False)
    acc, img = sc.accumulate(acc, rtmi.Opts(seed=5, tile_rows=8, tile_first=1, tile_stride=2, sample_first=7, sample_count=5))
    assert img.shape == whole.shape and np.array_equal(img, whole)


@pytest.mark.gpu
def test_cli_resumed_run_writes_the_same_ppm(tmp_path):
    one = tmp_path / "one.ppm"
    r = _run(["--rtiow", "-w", "80", "-h", "45", "-spp", "12", "-o", str(one), "--no-png"])
    assert r.returncode == 0, r.stderr
    part = tmp_path / "sums.bin"
    a = tmp_path / "a.ppm"
    r = _run(["--rtiow", "-w", "80", "-h", "45", "-spp", "5", "-o", str(a), "--no-png", "--acc-out", str(part)])
    assert r.returncode == 0, r.stderr
    b = tmp_path / "b.ppm"
    r = _run(["--rtiow", "-w", "80", "-h", "45", "-spp", "7", "-o", str(b), "--no-png", "--acc-in", str(part),
              "--acc-out", str(part)])
    assert r.returncode == 0 and "samples [5, 12)" in r.stderr, r.stderr
    assert one.read_bytes() == b.read_bytes()
    assert a.read_bytes() != b.read_bytes()
    hdr = struct.unpack("<8siiqQq", part.read_bytes()[:40])
    assert hdr == (b"RTMIACC2", 80, 45, 12, 2023, 24)
