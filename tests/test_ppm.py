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
