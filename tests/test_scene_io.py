"""Host logic: scene JSON -> tables (parser.hpp:504-573), builders, RTIOW generator, camera,
cylinder transforms, serialisation, error behaviour."""
import json
import math
import os

import numpy as np
import pytest


def test_three_sphere_tables(rtmi, scenes_dir):
    sc = rtmi.Scene.load(os.path.join(scenes_dir, "three_sphere.json"))
    i = sc.info
    assert (i.width, i.height, i.samples_per_pixel, i.max_depth) == (400, 225, 100, 50)
    assert (i.num_prims, i.num_materials, i.num_textures) == (5, 4, 2)
    assert i.flags == rtmi.FLAG_SKY_GRADIENT | rtmi.FLAG_DEFOCUS_BLUR
    p = sc.prims()
    assert list(p["type"]) == [0] * 5
    np.testing.assert_array_equal(p["f"][1][:4], np.float32([0, -100.5, -1, 100]))
    assert p["f"][4][3] == np.float32(-0.45)  # hollow glass: negative radius kept
    m = sc.materials()
    assert list(m["type"]) == [0, 0, 1, 2] and m["ir"][3] == np.float32(1.5)
    np.testing.assert_array_equal(sc.textures()["c0"][0], np.float32([0.1, 0.2, 0.5]))


def test_reference_scene_files_parse(rtmi, golden_dir):
    """The reference's own scene files (normalised copies under tests/golden/scenes)."""
    want = {"sample_scene": (6, 5, 4), "basic_scene": (0, 0, 0), "blue": (11, 8, 3), "blue2": (12, 9, 4)}
    for name, counts in want.items():
        sc = rtmi.Scene.load(os.path.join(golden_dir, "scenes", name + ".json"))
        i = sc.info
        assert (i.num_prims, i.num_materials, i.num_textures) == counts, name
        assert not (i.flags & rtmi.FLAG_SKY_GRADIENT)  # JSON scenes use the constant background
    sc = rtmi.Scene.load(os.path.join(golden_dir, "scenes", "blue.json"))
    assert (sc.width, sc.height, sc.spp) == (2560, 1440, 2000)
    t = list(sc.prims()["type"])
    assert t == [2, 1, 4, 4, 4, 4, 0, 0, 0, 0, 0]
    assert sc.materials()["fuzz"][0] == np.float32(0.5)


def test_reference_scene_files_parse_in_place(rtmi):
    """Where the reference checkout exists (this container), its files parse as shipped."""
    ref = "/root/reference/gpu-version"
    if not os.path.isdir(ref):
        pytest.skip("reference checkout not present")
    for name in ("sample_scene.json", "basic_scene.json", "blue.json", "blue2.json"):
        sc = rtmi.Scene.load(os.path.join(ref, name))
        raw = json.load(open(os.path.join(ref, name)))
        assert sc.info.num_prims == len(raw["object"]["data"])
        assert sc.width == raw["width"] and sc.spp == raw["samples_per_pixel"]


def test_json_round_trip_is_exact(rtmi, scenes_dir, golden_dir):
    for path in (os.path.join(scenes_dir, "mixed_emissive.json"), os.path.join(golden_dir, "scenes", "blue2.json")):
        a = rtmi.Scene.load(path)
        b = rtmi.Scene.parse(a.to_json())
        for ta, tb in ((a.prims(), b.prims()), (a.materials(), b.materials()), (a.textures(), b.textures())):
            assert ta.tobytes() == tb.tobytes()
        ca, cb = a.get_camera(), b.get_camera()
        assert bytes(ca) == bytes(cb)
        assert a.info.flags == b.info.flags and bytes(a.info) == bytes(b.info)


def test_metal_fuzz_is_clamped(rtmi):
    sc = rtmi.Scene.new(8, 8, 1)
    m = sc.metal((1, 1, 1), 3.0)  # material.cuh:61: fuzz(f < 1 ? f : 1)
    assert sc.materials()["fuzz"][m] == 1.0


@pytest.mark.parametrize("text,needle", [
    ("{", "JSON error"),
    ('{"a": [1, 2,, 3]}', "JSON error"),
    ("[]", "must be an object"),
    ('{"background": [0,0,0]}', "max_depth"),
])
def test_malformed_input_is_an_error(rtmi, text, needle):
    with pytest.raises(rtmi.RtmiError) as e:
        rtmi.Scene.parse(text)
    assert needle in str(e.value)


def _base():
    return {"background": [0, 0, 0], "max_depth": 5, "samples_per_pixel": 1, "width": 8, "height": 8,
            "camera": {"lookfrom": [0, 0, 5], "lookat": [0, 0, 0], "vup": [0, 1, 0], "vfov": 40, "aperture": 0},
            "object": {"data": []}, "material": {"data": []}, "texture": {"data": []}}


def test_scene_errors(rtmi):
    d = _base()
    d["object"]["data"].append({"type": "torus", "material": 0})
    with pytest.raises(rtmi.RtmiError, match="unknown type"):
        rtmi.Scene.parse(json.dumps(d))
    d = _base()
    d["object"]["data"].append({"type": "sphere", "center": [0, 0, 0], "radius": 1, "material": 0})
    with pytest.raises(rtmi.RtmiError, match="references material"):
        rtmi.Scene.parse(json.dumps(d))
    d = _base()
    d["material"]["data"].append({"type": "lambertian", "texture": 2})
    with pytest.raises(rtmi.RtmiError, match="references texture"):
        rtmi.Scene.parse(json.dumps(d))
    d = _base()
    d["width"] = 1
    with pytest.raises(rtmi.RtmiError, match=">= 2"):
        rtmi.Scene.parse(json.dumps(d))
    d = _base()
    del d["camera"]["vfov"]
    with pytest.raises(rtmi.RtmiError, match="vfov"):
        rtmi.Scene.parse(json.dumps(d))
    with pytest.raises(rtmi.RtmiError, match="cannot open"):
        rtmi.Scene.load("/nonexistent/scene.json")
    # an empty object list is a valid scene (basic_scene.json ships that way)
    assert rtmi.Scene.parse(json.dumps(_base())).info.num_prims == 0


def test_builder_equals_json(rtmi, scenes_dir):
    """camera/hittable/material constructor argument lists of the reference classes."""
    a = rtmi.Scene.load(os.path.join(scenes_dir, "three_sphere.json"))
    b = rtmi.Scene.new(400, 225, 100, 50)
    b.set_background((0.5, 0.7, 1.0), sky_gradient=True, defocus_blur=True)
    b.camera((-2, 2, 1), (0, 0, -1), (0, 1, 0), 20.0, 0.0, 0.0, 0.0)
    m0 = b.lambertian((0.1, 0.2, 0.5))
    m1 = b.lambertian((0.8, 0.8, 0.0))
    m2 = b.metal((0.8, 0.6, 0.2), 0.0)
    m3 = b.dielectric(1.5)
    b.sphere((0, 0, -1), 0.5, m0)
    b.sphere((0, -100.5, -1), 100, m1)
    b.sphere((1, 0, -1), 0.5, m2)
    b.sphere((-1, 0, -1), 0.5, m3)
    b.sphere((-1, 0, -1), -0.45, m3)
    assert a.prims().tobytes() == b.prims().tobytes()
    assert a.materials().tobytes() == b.materials().tobytes()
    assert a.textures().tobytes() == b.textures().tobytes()
    assert bytes(a.get_camera()) == bytes(b.get_camera())


def test_override_cli_flags(rtmi, scenes_dir):
    sc = rtmi.Scene.load(os.path.join(scenes_dir, "three_sphere.json"))
    sc.override(width=64, height=36, spp=7, max_depth=9)  # -w -h -spp -d, main.cpp:71-81
    i = sc.info
    assert (i.width, i.height, i.samples_per_pixel, i.max_depth) == (64, 36, 7, 9)
    sc.override()  # all zero: unchanged
    assert sc.info.width == 64
    with pytest.raises(rtmi.RtmiError):
        sc.override(width=1)
    assert sc.info.width == 64  # failed override leaves the scene as it was


def test_camera_matches_checker_derivation(rtmi, rtcheck, scenes_dir, golden_dir):
    """camera::camera (camera.cuh:9-29): product vs the checker's own fp64 derivation."""
    for path in (os.path.join(scenes_dir, "three_sphere.json"), os.path.join(golden_dir, "scenes", "sample_scene.json"),
                 os.path.join(golden_dir, "rtiow_seed7.json")):
        sc = rtmi.Scene.load(path)
        got = sc.get_camera()
        want = rtcheck.oracle_derive_camera(**rtcheck.camera_params_of(sc))
        for name in ("origin", "lower_left", "horizontal", "vertical", "u", "v", "w"):
            assert list(getattr(got, name)) == list(getattr(want, name)), (path, name)
        assert got.lens_radius == want.lens_radius


def test_camera_known_answer(rtmi):
    sc = rtmi.Scene.new(200, 100, 1)
    sc.camera((0, 0, 0), (0, 0, -1), (0, 1, 0), 90.0, 2.0, 0.0, 1.0)  # RTIOW's first camera
    c = sc.get_camera()
    np.testing.assert_allclose(list(c.horizontal), [4, 0, 0], atol=1e-6)
    np.testing.assert_allclose(list(c.vertical), [0, 2, 0], atol=1e-6)
    np.testing.assert_allclose(list(c.lower_left), [-2, -1, -1], atol=1e-6)


def test_cylinder_transform(rtmi):
    sc = rtmi.Scene.new(8, 8, 1)
    m = sc.lambertian((0.5, 0.5, 0.5))
    sc.cylinder(0.25, -1, 1, m, rotate=((0, 1, 0), 90), translate=(1, 2, 3))
    sc.cylinder(0.5, 0, 2, m, rotate=((0.3, 0.51, 1), 150))
    sc.cylinder(0.5, 0, 2, m, translate=(-1, 0, 4))
    sc.cylinder(0.5, 0, 2, m)
    p = sc.prims()

    def mat4(rows):
        return np.vstack([np.asarray(rows, dtype=np.float64).reshape(3, 4), [0, 0, 0, 1]])
    for k in range(4):
        M, Mi = mat4(p["m"][k]), mat4(p["m_inv"][k])
        np.testing.assert_allclose(M @ Mi, np.eye(4), atol=2e-6)
    # rotate about +y by 90 degrees then translate: the object z axis maps to world +x
    M = mat4(p["m"][0])
    np.testing.assert_allclose(M @ [0, 0, 1, 1], [1 + 1, 2, 3, 1], atol=1e-6)
    # Rodrigues rotation (vec3.cuh:396-418): the axis is a fixed point, angle preserved
    a = np.array([0.3, 0.51, 1.0]) / np.linalg.norm([0.3, 0.51, 1.0])
    R = mat4(p["m"][1])[:3, :3]
    np.testing.assert_allclose(R @ a, a, atol=1e-6)
    assert abs(np.trace(R) - (1 + 2 * math.cos(math.radians(150)))) < 1e-6
    np.testing.assert_array_equal(mat4(p["m"][3]), np.eye(4))
    with pytest.raises(rtmi.RtmiError, match="zero length"):
        sc.cylinder(0.5, 0, 2, m, rotate=((0, 0, 0), 10))


def test_rtiow_generator(rtmi):
    """random_scene(), main.cpp:125-172: structure, determinism, rejection rule."""
    a, b, c = rtmi.Scene.rtiow(7, 64, 36, 4), rtmi.Scene.rtiow(7, 64, 36, 4), rtmi.Scene.rtiow(8, 64, 36, 4)
    pa = a.prims()
    assert pa.tobytes() == b.prims().tobytes() and pa.tobytes() != c.prims().tobytes()
    n = len(pa)
    assert 470 <= n <= 488  # 1 ground + <= 484 small + 3 big
    np.testing.assert_array_equal(pa["f"][0][:4], np.float32([0, -1000, 0, 1000]))
    np.testing.assert_array_equal(pa["f"][-3][:4], np.float32([0, 1, 0, 1]))
    np.testing.assert_array_equal(pa["f"][-2][:4], np.float32([-4, 1, 0, 1]))
    np.testing.assert_array_equal(pa["f"][-1][:4], np.float32([4, 1, 0, 1]))
    small = pa["f"][1:-3]
    assert np.all(small[:, 1] == np.float32(0.2)) and np.all(small[:, 3] == np.float32(0.2))
    d = np.sqrt((small[:, 0].astype(np.float64) - 4) ** 2 + small[:, 2].astype(np.float64) ** 2)
    assert d.min() > 0.9  # main.cpp:139
    mats = a.materials()
    kinds = mats["type"][pa["material"][1:-3]]
    frac = [(kinds == k).mean() for k in (0, 1, 2)]
    assert 0.7 < frac[0] < 0.9 and 0.08 < frac[1] < 0.22 and 0.01 < frac[2] < 0.1
    metal = mats[mats["type"] == 1]
    assert metal["fuzz"][:-1].max() <= 0.5 and metal["albedo"][:-1].min() >= 0.5
    tex = a.textures()
    assert tex["type"][0] == 1  # checker ground: even (0.2,0.3,0.1), odd (0.9,0.9,0.9)
    np.testing.assert_array_equal(tex["c0"][0], np.float32([0.2, 0.3, 0.1]))
    np.testing.assert_array_equal(tex["c1"][0], np.float32([0.9, 0.9, 0.9]))
    i = a.info
    assert i.flags == 3
    cam = rtmi_cam = a.get_camera()
    assert list(cam.lookfrom) == [13, 2, 3] and abs(cam.lens_radius - 0.05) < 1e-9
    assert abs(cam.focus_dist - math.sqrt(13 * 13 + 4 + 9)) < 1e-5


def test_rtiow_matches_committed_fixture(rtmi, golden_dir):
    a = rtmi.Scene.rtiow(7, 48, 27, 8, 50)
    b = rtmi.Scene.load(os.path.join(golden_dir, "rtiow_seed7.json"))
    assert a.prims().tobytes() == b.prims().tobytes()
    assert a.materials().tobytes() == b.materials().tobytes()
    assert a.textures().tobytes() == b.textures().tobytes()


def test_aabb_slab(rtmi, rtcheck):
    """aabb::hit, aabb.hpp:15-29: known answers, product formula vs checker."""
    import ctypes as C
    lib = rtcheck.oracle_lib()
    cases = [
        ((-1, -1, -1), (1, 1, 1), (0, 0, 5), (0, 0, -1), 0.001, 1e30, True),
        ((-1, -1, -1), (1, 1, 1), (0, 0, 5), (0, 0, 1), 0.001, 1e30, False),   # pointing away
        ((-1, -1, -1), (1, 1, 1), (3, 0, 5), (0, 0, -1), 0.001, 1e30, False),  # misses in x
        ((-1, -1, -1), (1, 1, 1), (0, 0, 5), (0, 0, -1), 0.001, 3.0, False),   # t_max before the box
        ((-1, -1, -1), (1, 1, 1), (0, 0, 0), (1, 1, 1), 0.001, 1e30, True),    # origin inside
        ((0, 0, 0), (2, 2, 2), (-1, -1, -1), (1, 1, 1), 0.001, 1e30, True),    # through the diagonal
    ]
    f3 = lambda v: (C.c_float * 3)(*v)
    for bmin, bmax, o, d, t0, t1, want in cases:
        assert rtmi.aabb_hit(bmin, bmax, o, d, t0, t1) == want
        assert bool(lib.rto_aabb_hit(f3(bmin), f3(bmax), f3(o), f3(d), t0, t1)) == want
    rng = np.random.default_rng(3)
    for _ in range(500):
        lo = rng.uniform(-2, 0, 3); hi = lo + rng.uniform(0.1, 3, 3)
        o = rng.uniform(-5, 5, 3); d = rng.normal(size=3)
        a = rtmi.aabb_hit(lo, hi, o, d, 0.001, 100.0)
        assert a == bool(lib.rto_aabb_hit(f3(lo), f3(hi), f3(o), f3(d), 0.001, 100.0))
        # brute force: sample the ray densely
        ts = np.linspace(0.001, 100, 20001)
        pts = o[None, :] + ts[:, None] * d[None, :]
        inside = np.all((pts >= lo - 1e-6) & (pts <= hi + 1e-6), axis=1).any()
        if inside:
            assert a


def test_json_scenes_default_to_gpu_version_camera(rtmi, golden_dir):
    """The JSON interface is gpu-version's: its camera::get_ray has the lens sample disabled (camera.cuh:33-34) and
    misses return the constant background (main.cu:63).  A scene file without the two extension keys must select
    exactly that; the keys switch the cmake-cpu-version behaviour on."""
    import json
    d = json.load(open(os.path.join(golden_dir, "scenes", "sample_scene.json")))
    d.pop("defocus_blur", None), d.pop("sky_gradient", None)
    assert d["camera"]["aperture"] > 0  # the reference's files all carry an aperture the CUDA renderer ignores
    plain = rtmi.Scene.parse(json.dumps(d))
    assert plain.info.flags == 0
    d["defocus_blur"] = True
    assert rtmi.Scene.parse(json.dumps(d)).info.flags == rtmi.FLAG_DEFOCUS_BLUR
    d["sky_gradient"] = True
    assert rtmi.Scene.parse(json.dumps(d)).info.flags == rtmi.FLAG_DEFOCUS_BLUR | rtmi.FLAG_SKY_GRADIENT
    # the writer always states both, so a round trip never depends on the default
    again = rtmi.Scene.parse(plain.to_json())
    assert again.info.flags == 0 and '"defocus_blur": false' in plain.to_json()
    # the CPU-semantics entry points keep the lens: random_scene() and the constructors
    assert rtmi.Scene.rtiow(7, 32, 18, 1, 5).info.flags == rtmi.FLAG_DEFOCUS_BLUR | rtmi.FLAG_SKY_GRADIENT
    assert rtmi.Scene.new(16, 16, 1).info.flags == rtmi.FLAG_DEFOCUS_BLUR | rtmi.FLAG_SKY_GRADIENT
    assert rtmi.Scene.dna(0.0).info.flags == 0  # dna.py renders through gpu-version
