"""The animation harness (gpu-version/blue.py, blue2.py, dna.py) as library calls + rtmi-frames.
The Python scripts are never run or shipped; what they do is pinned by their text: every cylinder's
rotate.angle += step per frame (blue.py:16-19), the DNA formulas (dna.py:29-84), the output naming."""
import json
import math
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest


def _dna_expected(angle):
    """Independent restatement of dna.py:29-84 (objects / materials / textures of one frame)."""
    num_object, space = 5, 5
    tex, mat, obj = [], [], []
    for i, _ in enumerate(range(-num_object * 3, num_object * 3)):
        tex += [[232 / 256, 209 / 256, 209 / 256], [232 / 256, 209 / 256, 209 / 256], [202 / 256, 202 / 256, 224 / 256]]
        mat += [i * 3 + 0, i * 3 + 1, i * 3 + 2]
    for offset in range(3):
        for i, idv in enumerate(range(-num_object, num_object)):
            theta = (36 * (idv + num_object) + angle) / 180 * math.pi
            xo = offset * space - space
            zo = math.fabs(offset - 1) * -20 + 20
            obj.append(("sphere", [2.5 * math.cos(theta) + xo, idv, 2.5 * math.sin(theta) + zo], i * 3 + 0))
            obj.append(("sphere", [2.5 * math.cos(theta + math.pi) + xo, idv, 2.5 * math.sin(theta + math.pi) + zo], i * 3 + 1))
            obj.append(("cylinder", [xo, idv, zo], i * 3 + 2, 36 * -(idv + num_object) + 90 + angle))
    return tex, mat, obj


@pytest.mark.parametrize("angle", [0, 37, 359])
def test_dna_generator(rtmi, golden_dir, angle):
    sc = rtmi.Scene.dna(angle)
    i = sc.info
    assert (i.width, i.height, i.samples_per_pixel, i.max_depth) == (1600, 900, 100, 50)  # basic_scene.json
    assert (i.num_prims, i.num_materials, i.num_textures) == (90, 90, 90)
    tex, mat, obj = _dna_expected(angle)
    np.testing.assert_array_equal(sc.textures()["c0"], np.float32(tex))
    assert list(sc.materials()["texture"]) == mat and set(sc.materials()["type"]) == {3}
    p = sc.prims()
    d = json.loads(sc.to_json())["object"]["data"]
    for k, o in enumerate(obj):
        assert p["material"][k] == o[2]
        if o[0] == "sphere":
            assert p["type"][k] == 0 and p["f"][k][3] == 0.5
            np.testing.assert_array_equal(p["f"][k][:3], np.float32(o[1]))
        else:
            assert p["type"][k] == 4
            np.testing.assert_array_equal(p["f"][k][:3], np.float32([0.3, -2.18, 2.18]))
            assert d[k]["translate"] == [float(v) for v in o[1]] and d[k]["rotate"] == {"axis": [0, 1, 0], "angle": o[3]}
    # over a base scene: camera / size / background are the base's, objects are replaced
    base = rtmi.Scene.load(os.path.join(golden_dir, "scenes", "basic_scene.json"))
    base.override(width=320, height=180, spp=4)
    sc2 = rtmi.Scene.dna(angle, base)
    assert sc2.width == 320 and sc2.prims().tobytes() == p.tobytes()
    assert bytes(sc2.get_camera())[:40] == bytes(base.get_camera())[:40]


def test_rotate_cylinders_matches_json_edit(rtmi, golden_dir):
    """blue.py:16-19: item["rotate"]["angle"] += 1, cumulatively, for every cylinder."""
    path = os.path.join(golden_dir, "scenes", "blue.json")
    d = json.load(open(path))
    sc = rtmi.Scene.load(path)
    assert sc.rotate_cylinders(1.0) == 4
    assert sc.rotate_cylinders(1.0) == 4
    for o in d["object"]["data"]:
        if o["type"] == "cylinder":
            o["rotate"]["angle"] += 2
    want = rtmi.Scene.parse(json.dumps(d))
    assert sc.prims().tobytes() == want.prims().tobytes()
    # spheres / rects untouched, clone is independent
    c = sc.clone()
    c.rotate_cylinders(90)
    assert sc.prims().tobytes() == want.prims().tobytes() and c.prims().tobytes() != want.prims().tobytes()
    sc.set_output_file("./build/output/blue/frame_001.png")
    assert sc.output_file == "./build/output/blue/frame_001.png"


def _png_rgb(path):
    data = open(path, "rb").read()
    pos, idat, size = 8, b"", None
    while pos < len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if typ == b"IHDR":
            size = struct.unpack(">II", body[:8])
        if typ == b"IDAT":
            idat += body
        pos += 12 + n
    w, h = size
    return np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(h, 1 + 3 * w)[:, 1:].reshape(h, w, 3)


@pytest.mark.gpu
def test_rtmi_frames_driver(rtmi, rtcheck, golden_dir, tmp_path):
    exe = os.path.join(os.path.dirname(rtmi.LIB_PATH), "rtmi-frames")
    tmpl = os.path.join(golden_dir, "scenes", "blue2.json")
    out = str(tmp_path / "f_%03d.png")
    r = subprocess.run([exe, "--template", tmpl, "--frames", "3", "--step", "2", "--out", out, "--scene-out",
                        str(tmp_path / "s_%03d.json"), "-w", "64", "-h", "36", "-spp", "4", "--seed", "9"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    for k in range(3):
        sc = rtmi.Scene.load(tmpl)
        sc.override(width=64, height=36, spp=4)
        sc.rotate_cylinders(2.0 * (k + 1))  # blue2.py bumps the angle before writing frame k
        want = rtmi.quantize_rgb8(sc.render(rtmi.Opts(seed=9)), 4, gamma=False)
        np.testing.assert_array_equal(_png_rgb(out % k), want)
        if k == 1:  # ... and one frame against the CPU checker's pixels (not only the library against itself)
            ref, _ = rtcheck.oracle_render(sc, seed=9)
            np.testing.assert_array_equal(_png_rgb(out % k), rtmi.quantize_rgb8(ref, 4, gamma=False))
        dumped = rtmi.Scene.load(str(tmp_path / ("s_%03d.json" % k)))
        assert dumped.prims().tobytes() == sc.prims().tobytes() and dumped.output_file == out % k
    # the DNA animation, two frames
    r = subprocess.run([exe, "--dna", "--frames", "2", "--first", "5", "--out", str(tmp_path / "d_%03d.png"), "-w", "48",
                        "-h", "27", "-spp", "2"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    sc = rtmi.Scene.dna(6)
    sc.override(width=48, height=27, spp=2)
    np.testing.assert_array_equal(_png_rgb(str(tmp_path / "d_006.png")), rtmi.quantize_rgb8(sc.render(), 2, gamma=False))
    ref, _ = rtcheck.oracle_render(sc, seed=rtmi.Opts().seed)
    np.testing.assert_array_equal(_png_rgb(str(tmp_path / "d_006.png")), rtmi.quantize_rgb8(ref, 2, gamma=False))
    assert "total time" in r.stderr
