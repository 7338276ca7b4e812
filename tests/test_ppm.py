"""output_image / write_color (main.cu:359-372, color.cuh:70-95; cpu color.h:14-35)."""
import os

import numpy as np


def test_ppm_bytes(rtmi, tmp_path):
    img = np.zeros((2, 3, 3), dtype=np.float32)  # 3 wide, 2 high, row 0 = bottom
    spp = 4
    img[0, 0] = [4.0, 0.0, 1.0]      # bottom-left: mean (1, 0, 0.25) -> 255, 0, 128
    img[1, 2] = [0.16, 4e6, 0.04]    # top-right
    path = str(tmp_path / "out.ppm")
    rtmi.output_image(img, spp, path)
    text = open(path).read()
    lines = text.split("\n")
    assert lines[0] == "P3" and lines[1] == "3 2" and lines[2] == "255"
    body = lines[3:-1]
    assert len(body) == 6 and text.endswith("\n")
    # rows are written top to bottom (j = H-1 .. 0)
    assert body[2] == f"{int(256 * np.sqrt(np.float32(0.04)))} 255 {int(256 * np.sqrt(np.float32(0.01)))}"
    assert body[3] == "255 0 128"
    assert body[0] == "0 0 0"


def test_quantisation_matches_reference_write_color(rtmi, rtcheck, golden_dir):
    """Golden outputs of the reference's own write_color (fp64) on a sweep of sums; the fp32
    writer may differ only where sqrt(sum/spp)*256 sits within float rounding of an integer."""
    z = np.load(os.path.join(golden_dir, "ref_write_color.npz"))
    spp = int(z["spp"])
    sums = z["sums"]
    img = np.zeros((1, len(sums), 3), dtype=np.float32)
    img[0, :, 0], img[0, :, 1], img[0, :, 2] = sums, sums / 2, sums / 3
    got = rtmi.quantize_rgb8(img, spp)[0].astype(np.int32)
    want = z["out"]
    assert np.abs(got - want).max() <= 1
    assert (got != want).mean() < 0.01
    lib = rtcheck.oracle_lib()
    chk = np.array([[lib.rto_quantize(float(v), spp, 1) for v in px] for px in img[0]])
    np.testing.assert_array_equal(got, chk)
    if rtcheck.have_ref():
        live = np.array([rtcheck.ref_write_color([s, s / 2, s / 3], spp) for s in sums])
        np.testing.assert_array_equal(live, want)


def test_linear_png_bytes(rtmi, rtcheck):
    """write_image (color.cuh:15-35): no gamma."""
    img = np.float32([[[0.5 * 8, 2.0 * 8, 0.25 * 8]]])
    out = rtmi.quantize_rgb8(img, 8, gamma=False)[0, 0]
    assert list(out) == [128, 255, 64]
    lib = rtcheck.oracle_lib()
    assert [lib.rto_quantize(float(v), 8, 0) for v in img[0, 0]] == [128, 255, 64]


def test_png_writer_round_trip(rtmi, tmp_path):
    """rt_write_png: a valid PNG (signature, IHDR, zlib stream, CRCs) holding write_image's bytes."""
    import struct
    import zlib
    rng = np.random.default_rng(0)
    img = rng.uniform(0, 3, (37, 301, 3)).astype(np.float32) * 5  # > 65535 raw bytes: several stored blocks
    path = str(tmp_path / "o.png")
    rtmi.write_image(img, 5, path)
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, []
    while pos < len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        (crc,) = struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])
        assert zlib.crc32(typ + body) == crc
        chunks.append((typ, body))
        pos += 12 + n
    assert [c[0] for c in chunks] == [b"IHDR", b"IDAT", b"IEND"]
    w, h, depth, ctype = struct.unpack(">IIBB", chunks[0][1][:10])
    assert (w, h, depth, ctype) == (301, 37, 8, 2)
    raw = zlib.decompress(chunks[1][1])
    rows = np.frombuffer(raw, dtype=np.uint8).reshape(37, 1 + 301 * 3)
    assert not rows[:, 0].any()
    np.testing.assert_array_equal(rows[:, 1:].reshape(37, 301, 3), rtmi.quantize_rgb8(img, 5, gamma=False))


def test_fixed_point_conversion_without_fp64(tmp_path):
    """The kernel converts a sample to 64-bit fixed point without fp64 (trunc + exact fraction + v_rndne);
    tests/fixed_point_check.c restates that expression in C and compares it with the checker's
    llrint((double)v * 2^32) on every 37th fp32 bit pattern (all 2^32 take 20 s: run it with stride 1)."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "fixed_point_check")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-o", exe, os.path.join(root, "tests", "fixed_point_check.c"), "-lm"],
                   check=True)
    r = subprocess.run([exe, "37"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout
    assert " 0 mismatches" in r.stdout
