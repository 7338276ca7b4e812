"""The Taichi renderer's extras: image textures read at the hit record's (u, v)
(taichi-version/material.py:137-144), the (u, v) of every primitive (gpu-version/object.cuh:87-93, 113-114, 283-288),
triangles (taichi-version/hittable.py:38-71, 95-110) and the OBJ reader (taichi-version/main.py:23-41, 110-118).

PARITY UNPINNED: Taichi is not installed here and nvcc is absent, so no reference run exists for these features.  The CPU
checker restates them from the sources; this file holds closed-form known-answer tests of that restatement (CPU), and
the HIP path is held to bit-equality with it (-m gpu)."""
import json
import math
import os

import numpy as np
import pytest

SEED = 2023


def _texture_pixels(rows=6, cols=5):
    rng = np.random.default_rng(3)
    return rng.integers(0, 256, size=(rows, cols, 3), dtype=np.uint8)


def _probe(rtmi, bg=(0, 0, 0), lookfrom=(0, 0, 5), lookat=(0, 0, 0), vfov=10.0, depth=50, size=9):
    sc = rtmi.Scene.new(size, size, 4, depth)
    sc.set_background(bg, sky_gradient=False, defocus_blur=False)
    sc.camera(lookfrom, lookat, (0, 1, 0), vfov, 1.0, 0.0, 1.0)
    return sc


# ---------------------------------------------------------------- CPU: the checker's restatement
def test_fixed_sequence_trig_is_accurate(rtcheck):
    lib = rtcheck.oracle_lib()
    import ctypes as C
    lib.rto_atan2f.restype = C.c_float
    lib.rto_atan2f.argtypes = [C.c_float, C.c_float]
    lib.rto_acosf.restype = C.c_float
    lib.rto_acosf.argtypes = [C.c_float]
    rng = np.random.default_rng(1)
    worst = 0.0
    for y, x in np.float32(rng.uniform(-3, 3, size=(4000, 2))):
        worst = max(worst, abs(lib.rto_atan2f(float(y), float(x)) - math.atan2(float(y), float(x))))
    for y, x in ((0.0, 1.0), (1.0, 0.0), (0.0, -1.0), (-1.0, 0.0), (1.0, 1.0), (-1.0, -1.0), (1e-20, 1.0), (1.0, 1e-20)):
        worst = max(worst, abs(lib.rto_atan2f(y, x) - math.atan2(y, x)))
    assert worst < 6e-7  # ~2 ulp of pi
    assert lib.rto_atan2f(0.0, 0.0) == 0.0
    worst = max(abs(lib.rto_acosf(float(c)) - math.acos(float(c))) for c in np.float32(np.linspace(-1, 1, 2001)))
    assert worst < 1e-6


def test_sphere_rect_cylinder_uv_known_answers(rtmi, rtcheck):
    sc = _probe(rtmi)
    m = sc.lambertian((0.5, 0.5, 0.5))
    sc.sphere((0, 0, 0), 1.0, m)
    osc = rtcheck.OracleScene(sc)
    # get_sphere_uv (object.cuh:87-93): theta = acos(-y), phi = atan2(-z, x) + pi; u = phi / 2pi, v = theta / pi
    for o, want in (((5, 0, 0), (0.5, 0.5)),      # +x pole: phi = pi
                    ((0, 5, 0), (None, 1.0)),      # top: theta = pi
                    ((0, -5, 0), (None, 0.0)),     # bottom
                    ((0, 0, 5), (0.25, 0.5)),      # +z: atan2(-1, 0) = -pi/2 -> phi = pi/2
                    ((0, 0, -5), (0.75, 0.5)),
                    ((-5, 0, 1e-6), (1.0, 0.5))):  # -x seen from slightly +z: atan2(-0, -1) -> phi -> 2 pi (u -> 1, or 0)
        hit, (u, v), t, prim = rtcheck.oracle_hit_uv(osc, o, [-c for c in o])
        assert hit and prim == 0 and abs(t - 0.8) < 1e-5
        if want[0] is not None:
            assert min(abs(u - want[0]), abs(u - want[0] + 1), abs(u - want[0] - 1)) < 2e-6
        assert abs(v - want[1]) < 2e-6
    # rects: (x - x0) / (x1 - x0), (y - y0) / (y1 - y0)   (object.cuh:113-114)
    sc = _probe(rtmi)
    m = sc.lambertian((0.5, 0.5, 0.5))
    sc.xy_rect(-1, 3, -2, 2, 0.0, m)
    sc.xz_rect(10, 12, 0, 4, 1.0, m)
    sc.yz_rect(-1, 1, 20, 24, -3.0, m)
    osc = rtcheck.OracleScene(sc)
    for o, d, prim, want in (((0, 1.5, 5), (0, 0, -1), 0, (0.25, 0.875)), ((11.5, 9, 1), (0, -1, 0), 1, (0.75, 0.25)),
                             ((4, 0.5, 23), (-1, 0, 0), 2, (0.75, 0.75))):
        hit, uv, t, p = rtcheck.oracle_hit_uv(osc, o, d)
        assert hit and p == prim and np.allclose(uv, want, atol=1e-6)
    # cylinder (object.cuh:283-288): u = (atan2(y, x) + 2 pi) / 4 pi, v = (z - zmin) / (zmax - zmin) in object space
    sc = _probe(rtmi)
    m = sc.lambertian((0.5, 0.5, 0.5))
    sc.cylinder(1.0, -1.0, 3.0, m)  # about z
    osc = rtcheck.OracleScene(sc)
    for o, d, want in (((5, 0, 0), (-1, 0, 0), (0.5, 0.25)), ((0, 5, 2), (0, -1, 0), (0.625, 0.75)),
                       ((0, -5, 1), (0, 1, 0), (0.375, 0.5))):
        hit, uv, t, p = rtcheck.oracle_hit_uv(osc, o, d)
        assert hit and abs(t - 4.0) < 1e-5 and np.allclose(uv, want, atol=1e-6)


def test_triangle_known_answers(rtmi, rtcheck):
    """Plane hit + four same-side tests + area weights (hittable.py:38-71); uv = u1 w1 + u2 w2 + u3 w3 with
    w1 = |(r-v1)x(r-v2)| / |(v3-v1)x(v3-v2)| -- the weight of the corner OPPOSITE to the edge v1 v2, as the reference has it."""
    v1, v2, v3 = np.float64([0, 0, 0]), np.float64([4, 0, 0]), np.float64([0, 2, 0])
    u1, u2, u3 = (0.0, 0.0), (1.0, 0.0), (0.0, 1.0)
    sc = _probe(rtmi)
    sc.triangle(v1, v2, v3, sc.lambertian((0.5, 0.5, 0.5)), u1, u2, u3)
    p = sc.prims()[0]
    assert p["type"] == rtmi.PRIM_TRIANGLE and np.allclose(p["m"][9:12], [0, 0, 1])
    osc = rtcheck.OracleScene(sc)
    rng = np.random.default_rng(5)
    for _ in range(200):
        a, b = rng.uniform(0.02, 0.96, 2)
        if a + b > 0.98:
            continue
        g = 1 - a - b
        target = a * v1 + b * v2 + g * v3
        origin = target + np.float64([rng.uniform(-1, 1), rng.uniform(-1, 1), rng.choice([-3.0, 2.0])])
        hit, uv, t, prim = rtcheck.oracle_hit_uv(osc, origin, target - origin)
        assert hit and abs(t - 1.0) < 1e-5  # the parameter of the un-normalised direction
        # w1 = weight of v3 (= g), w2 = weight of v2 (= b), w3 = weight of v1 (= a)
        want = np.float64(u1) * g + np.float64(u2) * b + np.float64(u3) * a
        assert np.allclose(uv, want, atol=2e-5)
    # outside, parallel and behind
    for o, d in (((5, 5, 3), (0, 0, -1)), ((-0.1, 1, 3), (0, 0, -1)), ((1, 1, 3), (1, 0, 0)), ((1, 0.5, 3), (0, 0, 1))):
        assert not rtcheck.oracle_hit_uv(osc, o, d)[0]
    # the hit normal is turned against the ray on both sides: a white sky seen through a mirror triangle
    for z in (3.0, -3.0):
        sc = _probe(rtmi, bg=(1, 1, 1), lookfrom=(1, 0.5, z), lookat=(1, 0.5, 0))
        sc.triangle(v1, v2, v3, sc.metal((0.5, 0.25, 1.0), 0.0))
        rgb, q = rtcheck.oracle_sample(sc, 5, 4, 4, 0)
        assert q == 2 and np.allclose(rgb, [0.5, 0.25, 1.0])
    with pytest.raises(rtmi.RtmiError, match="zero area"):
        sc.triangle((0, 0, 0), (1, 1, 1), (2, 2, 2), 0)


def test_image_texture_lookup(rtmi, rtcheck):
    """texel[int(frac(u) * rows)][int(frac(v) * cols)] / 255 at the hit's (u, v) (material.py:137-144): an emissive
    rect showing a 2 x 2 image -- every pixel well inside a quadrant is exactly that texel."""
    px = np.uint8([[[255, 0, 0], [0, 255, 0]], [[0, 0, 255], [51, 102, 204]]])  # [row (from u)][col (from v)]
    sc = _probe(rtmi, vfov=30.0)
    tex = sc.image_texture(px)
    assert np.array_equal(sc.get_image(tex), px)
    t = sc.textures()[tex]
    assert t["type"] == rtmi.TEX_IMAGE and list(t["c0"]) == [0.0, 2.0, 2.0]
    sc.xy_rect(-2, 2, -2, 2, 0.0, sc.diffuse_light(tex))
    # pixel (x, y) of the 9 x 9 view: u = (hit x + 2) / 4 grows with x, v with y (row 0 = bottom)
    for (x, y), (r, c) in (((2, 2), (0, 0)), ((2, 6), (0, 1)), ((6, 2), (1, 0)), ((6, 6), (1, 1))):
        rgb, _ = rtcheck.oracle_sample(sc, 5, x, y, 0)
        assert np.array_equal(rgb, px[r, c].astype(np.float32) / np.float32(255.0))
    # u = 1 exactly (the rect's far edge) indexes row `rows`: clamped to the last texel instead of reading outside
    hit, uv, _, _ = rtcheck.oracle_hit_uv(sc, (2.0, 0.5, 5), (0, 0, -1))
    assert hit and uv[0] == 1.0
    with pytest.raises(rtmi.RtmiError):
        sc.image_texture(np.zeros((0, 4, 3), dtype=np.uint8))


def test_scene_files_with_images_triangles_and_meshes(rtmi, tmp_path):
    px = _texture_pixels()
    ppm6 = tmp_path / "tex6.ppm"
    ppm6.write_bytes(b"P6\n# a comment\n%d %d\n255\n" % (px.shape[1], px.shape[0]) + px.tobytes())
    ppm3 = tmp_path / "tex3.ppm"
    ppm3.write_text("P3\n%d %d\n255\n" % (px.shape[1], px.shape[0]) + " ".join(str(v) for v in px.reshape(-1)) + "\n")
    obj = tmp_path / "quad.obj"
    obj.write_text("# a unit quad\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nf 1 2 3\nf 1/1 3/3/1 4/4 \n")
    scene = {
        "background": [0.1, 0.2, 0.3], "max_depth": 5, "samples_per_pixel": 2, "width": 16, "height": 12,
        "camera": {"lookfrom": [0, 0, 5], "lookat": [0, 0, 0], "vup": [0, 1, 0], "vfov": 40, "aperture": 0.0},
        "texture": {"data": [{"type": "image", "file": "tex6.ppm"}, {"type": "image", "file": str(ppm3)},
                             {"type": "image", "rows": 1, "cols": 2, "data": [1, 2, 3, 4, 5, 6]},
                             {"type": "solid_color", "color": [0.5, 0.5, 0.5]}]},
        "material": {"data": [{"type": "lambertian", "texture": 0}, {"type": "diffuse_light", "texture": 2},
                              {"type": "lambertian", "texture": 3}]},
        "object": {"data": [
            {"type": "triangle", "v1": [0, 0, 0], "v2": [1, 0, 0], "v3": [0, 1, 0], "u1": [0, 0], "u2": [1, 0], "u3": [0, 1],
             "material": 0},
            {"type": "mesh", "file": "quad.obj", "material": 1, "scale": 2.0, "matrix": [0, 0, 1, 0, 1, 0, 1, 0, 0],
             "translate": [4.0, 1.0, 2.0]},
            {"type": "sphere", "center": [0, 0, -3], "radius": 1.0, "material": 2}]},
    }
    f = tmp_path / "scene.json"
    f.write_text(json.dumps(scene))
    sc = rtmi.Scene.load(str(f))  # "file" entries are relative to the scene file
    prims, texs = sc.prims(), sc.textures()
    assert [int(t) for t in prims["type"]] == [5, 5, 5, 0] and [int(t) for t in texs["type"]] == [2, 2, 2, 0]
    assert np.array_equal(sc.get_image(0), px) and np.array_equal(sc.get_image(1), px)
    assert np.array_equal(sc.get_image(2), np.uint8([[[1, 2, 3], [4, 5, 6]]]))
    # the mesh: scale * (M v) + translate with M = the reference's axis swap (main.py:112-118): (x, y, z) -> (z, y, x)
    assert np.allclose(prims[1]["m"][:9], [4, 1, 2, 4, 1, 4, 4, 3, 4])
    assert np.allclose(prims[2]["m"][:9], [4, 1, 2, 4, 3, 4, 4, 3, 2])
    assert np.allclose(prims[1]["m_inv"][:6], [0, 0, 1, 0, 1, 1]) and np.allclose(prims[2]["m_inv"][:6], [0, 0, 1, 1, 0, 1])
    # round trip: triangles are written out, images by file name or inline
    again = rtmi.Scene.parse(sc.to_json().replace('"tex6.ppm"', json.dumps(str(ppm6))))
    assert np.array_equal(again.prims(), prims) and np.array_equal(again.textures(), texs)
    assert np.array_equal(again.get_image(2), sc.get_image(2))
    # through the constructors
    b = rtmi.Scene.new(16, 12, 2, 5)
    assert b.add_obj(str(obj), b.lambertian((0.5, 0.5, 0.5)), scale=2.0, matrix=[[0, 0, 1], [0, 1, 0], [1, 0, 0]],
                     translate=(4, 1, 2)) == 2
    assert np.array_equal(b.prims()["m"], prims[1:3]["m"])
    assert b.image_texture(str(ppm3)) == 1  # texture 0 is the lambertian's solid colour
    for bad in ('{"type": "image", "rows": 1, "cols": 1, "data": [1, 2]}', '{"type": "image", "file": "missing.ppm"}'):
        d = dict(scene)
        d["texture"] = {"data": [json.loads(bad)] + scene["texture"]["data"][1:]}
        (tmp_path / "bad.json").write_text(json.dumps(d))
        with pytest.raises(rtmi.RtmiError):
            rtmi.Scene.load(str(tmp_path / "bad.json"))
    (tmp_path / "bad.obj").write_text("v 0 0 0\nv 1 0 0\nf 1 2 9\n")
    with pytest.raises(rtmi.RtmiError, match="vertex 9"):
        b.add_obj(str(tmp_path / "bad.obj"), 0)


# ---------------------------------------------------------------- GPU: the HIP path == the checker
def _showcase(rtmi, tmp_path, w=72, h=45, spp=6):
    """Every primitive wearing an image texture, a triangle mesh, and the rest of the material zoo."""
    sc = rtmi.Scene.new(w, h, spp, 12)
    sc.set_background((0.55, 0.65, 0.9), sky_gradient=False, defocus_blur=True)
    sc.camera((6, 4, 9), (0, 0.8, 0), (0, 1, 0), 35.0, 0.0, 0.05, 0.0)
    img = sc.image_texture(_texture_pixels(16, 12))
    img2 = sc.image_texture(_texture_pixels(3, 7))
    lam_img, light_img = sc.lambertian(img), sc.diffuse_light(img2)
    grey, glass, metal = sc.lambertian((0.6, 0.6, 0.6)), sc.dielectric(1.5), sc.metal((0.8, 0.7, 0.6), 0.1)
    chk = sc.lambertian(sc.checker_texture((0.2, 0.3, 0.1), (0.9, 0.9, 0.9)))
    sc.xz_rect(-8, 8, -8, 8, 0.0, chk)
    sc.sphere((0, 1, 0), 1.0, lam_img)
    sc.sphere((-2.5, 0.7, 1.0), 0.7, glass)
    sc.sphere((2.2, 0.5, 2.0), 0.5, metal)
    sc.xy_rect(-3, 3, 0.5, 3.0, -3.0, light_img)
    sc.yz_rect(0.2, 2.5, -2, 2, -4.0, lam_img)
    sc.cylinder(0.5, -1.0, 1.0, lam_img, rotate=((1, 0, 0), 90.0), translate=(3.5, 1.0, -1.0))
    sc.cylinder(0.3, 0.0, 1.5, light_img, rotate=((0, 1, 1), 30.0), translate=(-3.5, 0.2, 2.0))
    obj = tmp_path / "roof.obj"
    obj.write_text("v -1 0 -1\nv 1 0 -1\nv 1 0 1\nv -1 0 1\nv 0 1.2 0\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nvt 0.5 0.5\n"
                   "f 1 2 5\nf 2 3 5\nf 3 4 5\nf 4 1 5\n")
    assert sc.add_obj(str(obj), lam_img, scale=0.8, translate=(-0.5, 2.0, 3.0)) == 4
    sc.triangle((1, 0.01, 4), (3, 0.01, 4), (2, 1.5, 3.5), metal)
    sc.triangle((-4, 0.2, -1), (-2, 0.2, -2), (-3, 2.2, -1.5), light_img, (0, 0), (1, 0), (0.5, 1))
    sc.triangle((3.2, 0.1, 0.5), (4.5, 0.1, 1.5), (3.8, 1.4, 1.0), glass)
    return sc


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [0, 16, 36, 44])
def test_textured_scene_bit_exact(rtmi, rtcheck, tmp_path, variant):
    sc = _showcase(rtmi, tmp_path)
    img = sc.render(rtmi.Opts(seed=SEED, variant=variant))
    ref, _ = rtcheck.oracle_render(sc, seed=SEED)
    assert np.abs(img - ref).max() / sc.spp < 1e-3
    assert np.array_equal(img, ref), f"{(img != ref).any(axis=2).sum()} pixels differ from the CPU checker"
    assert img.std() > 0.05 * sc.spp  # the frame shows something


@pytest.mark.gpu
def test_textured_scene_shards_ranges_and_limits(rtmi, rtcheck, tmp_path):
    sc = _showcase(rtmi, tmp_path, w=130, h=33, spp=5)
    full = sc.render(rtmi.Opts(seed=7))
    ref, _ = rtcheck.oracle_render(sc, seed=7)
    assert np.array_equal(full, ref)
    out = np.zeros_like(full)
    for r in range(3):
        o = rtmi.Opts(seed=7, tile_rows=4, tile_first=r, tile_stride=3)
        sc.scatter_rows(o, sc.render(o), out)
    assert np.array_equal(out, full)
    part = sc.render(rtmi.Opts(seed=7, sample_first=2, sample_count=3, spp_chunk=2))
    ref, _ = rtcheck.oracle_render(sc, seed=7, sample_first=2, sample_count=3)
    assert np.array_equal(part, ref)
    # the cluster searches are not built with triangles / image textures; the counting kernel is (round 3)
    with pytest.raises(rtmi.RtmiError) as e:
        sc.render(rtmi.Opts(variant=64))
    assert e.value.status == 6 and "triangle" in str(e.value)
    st, img = sc.count(rtmi.Opts(seed=7), want_image=True)
    assert np.array_equal(img, full) and st.kernel_variant == 36 and st.samples == 130 * 33 * 5


@pytest.mark.gpu
def test_many_triangles_are_culled_by_their_boxes(rtmi, rtcheck):
    """A 12 x 12 height-field mesh (288 triangles) over a textured ground: the culled kernel, the flat scan and the
    checker agree bit for bit."""
    sc = rtmi.Scene.new(64, 40, 3, 8)
    sc.set_background((0.7, 0.8, 1.0), sky_gradient=True, defocus_blur=False)
    sc.camera((7, 6, 9), (0, 0.5, 0), (0, 1, 0), 40.0)
    rng = np.random.default_rng(4)
    img = sc.lambertian(sc.image_texture(_texture_pixels(8, 8)))
    mats = [img, sc.metal((0.8, 0.8, 0.8), 0.05), sc.lambertian((0.7, 0.3, 0.3))]
    n = 12
    hgt = rng.uniform(0.0, 1.0, size=(n + 1, n + 1))
    P = lambda i, j: (i * 0.5 - 3.0, float(hgt[i, j]), j * 0.5 - 3.0)
    U = lambda i, j: (i / n, j / n)
    for i in range(n):
        for j in range(n):
            m = mats[(i + j) % 3]
            sc.triangle(P(i, j), P(i + 1, j), P(i, j + 1), m, U(i, j), U(i + 1, j), U(i, j + 1))
            sc.triangle(P(i + 1, j), P(i + 1, j + 1), P(i, j + 1), m, U(i + 1, j), U(i + 1, j + 1), U(i, j + 1))
    sc.sphere((0, -1000, 0), 1000.0, img)
    a = sc.render(rtmi.Opts(seed=SEED))
    assert np.array_equal(a, sc.render(rtmi.Opts(seed=SEED, variant=16)))
    ref, _ = rtcheck.oracle_render(sc, seed=SEED)
    assert np.array_equal(a, ref)


def _png_bytes(px, color_type, filters, level=9, palette=None):
    """A PNG file made by hand (zlib from the standard library): px = (rows, cols, channels) uint8."""
    import struct, zlib

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)

    rows, cols, ch = px.shape
    raw = bytearray()
    prev = np.zeros(cols * ch, dtype=np.int32)
    for y in range(rows):
        cur = px[y].reshape(-1).astype(np.int32)
        f = filters[y % len(filters)]
        a = np.concatenate([np.zeros(ch, dtype=np.int32), cur[:-ch]])
        c = np.concatenate([np.zeros(ch, dtype=np.int32), prev[:-ch]])
        if f == 0: pred = 0 * cur
        elif f == 1: pred = a
        elif f == 2: pred = prev
        elif f == 3: pred = (a + prev) // 2
        else:
            p = a + prev - c
            pa, pb, pc = abs(p - a), abs(p - prev), abs(p - c)
            pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, c))
        raw.append(f)
        raw += bytes(((cur - pred) & 255).astype(np.uint8))
        prev = cur
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", cols, rows, 8, color_type, 0, 0, 0))
    if palette is not None:
        out += chunk(b"PLTE", bytes(np.asarray(palette, dtype=np.uint8).reshape(-1)))
    z = zlib.compress(bytes(raw), level)
    out += chunk(b"IDAT", z[: len(z) // 2]) + chunk(b"IDAT", z[len(z) // 2:])  # split over two IDAT chunks
    return out + chunk(b"IEND", b"")


def test_png_textures(rtmi, tmp_path):
    rng = np.random.default_rng(8)
    smooth = (np.add.outer(np.arange(40) * 5, np.arange(33) * 3)[:, :, None] + np.arange(4)[None, None, :] * 40) % 256
    noisy = rng.integers(0, 256, size=(17, 23, 4))
    for name, base in (("smooth", smooth), ("noisy", noisy)):
        base = base.astype(np.uint8)
        for ctype, ch in ((2, 3), (6, 4), (0, 1), (4, 2)):
            px = np.ascontiguousarray(base[:, :, :ch])
            for level in (0, 1, 9):  # stored blocks, fixed and dynamic Huffman codes
                f = tmp_path / f"{name}_{ctype}_{level}.png"
                f.write_bytes(_png_bytes(px, ctype, filters=[0, 1, 2, 3, 4], level=level))
                sc = rtmi.Scene.new(16, 16, 1)
                tex = sc.image_texture(str(f))
                want = px[:, :, :3] if ch >= 3 else np.repeat(px[:, :, :1], 3, axis=2)
                assert np.array_equal(sc.get_image(tex), want), (name, ctype, level)
    # palette
    idx = rng.integers(0, 5, size=(9, 11, 1)).astype(np.uint8)
    pal = rng.integers(0, 256, size=(5, 3)).astype(np.uint8)
    f = tmp_path / "pal.png"
    f.write_bytes(_png_bytes(idx, 3, filters=[0], palette=pal))
    sc = rtmi.Scene.new(16, 16, 1)
    assert np.array_equal(sc.get_image(sc.image_texture(str(f))), pal[idx[:, :, 0]])
    # a JPEG called .png (what the reference's asset/tex/bricks2.png is) and a truncated PNG are errors, not garbage
    (tmp_path / "fake.png").write_bytes(b"\xff\xd8\xff\xe0" + b"\0" * 64)
    with pytest.raises(rtmi.RtmiError):
        sc.image_texture(str(tmp_path / "fake.png"))
    good = _png_bytes(smooth[:, :, :3].astype(np.uint8), 2, filters=[4])
    (tmp_path / "cut.png").write_bytes(good[: len(good) // 2])
    with pytest.raises(rtmi.RtmiError):
        sc.image_texture(str(tmp_path / "cut.png"))
    # a deflate bomb: a header that says 4 x 4 in front of 8 MB of zeros (8 KB compressed): refused as soon as the stream
    # outgrows the image it belongs to, not inflated first
    import struct, zlib
    chunk = lambda tag, data: struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    bomb = (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 4, 4, 8, 2, 0, 0, 0))
            + chunk(b"IDAT", zlib.compress(bytes(8 << 20), 9)) + chunk(b"IEND", b""))
    (tmp_path / "bomb.png").write_bytes(bomb)
    with pytest.raises(rtmi.RtmiError, match="expands beyond"):
        sc.image_texture(str(tmp_path / "bomb.png"))
