"""Analytic known-answer tests of the checker for the features that exist only in the
reference's CUDA code (no runnable reference here: "parity unpinned"): rects, cylinders +
transforms, emission, constant background, checker texture; plus generic integrator laws.
Each scene is built through the product's builder API and evaluated by the CPU checker
(the same scenes run on the HIP path in test_gpu_parity.py)."""
import math

import numpy as np


def _probe_scene(rtmi, bg=(0, 0, 0), sky=False, depth=50, lookfrom=(0, 0, 5), lookat=(0, 0, 0), vfov=10.0):
    sc = rtmi.Scene.new(9, 9, 4, depth)
    sc.set_background(bg, sky_gradient=sky, defocus_blur=False)
    sc.camera(lookfrom, lookat, (0, 1, 0), vfov, 1.0, 0.0, 1.0)
    return sc


def _center_px(rtcheck, sc, n=64, x=4, y=4):
    vals = np.array([rtcheck.oracle_sample(sc, 5, x, y, s)[0] for s in range(n)], dtype=np.float64)
    return vals


def test_furnace_emissive_rect_is_exact(rtmi, rtcheck):
    """A diffuse_light seen directly returns exactly its emission (main.cu:48-58)."""
    sc = _probe_scene(rtmi, bg=(0.25, 0.5, 0.75))
    light = sc.diffuse_light((2.0, 3.0, 4.0))
    sc.xy_rect(-1, 1, -1, 1, 0.0, light)
    v = _center_px(rtcheck, sc)
    assert np.all(v == np.float32([2, 3, 4]))
    # a corner pixel of a wider view misses the rect -> exactly the constant background
    sc2 = _probe_scene(rtmi, bg=(0.25, 0.5, 0.75), vfov=90.0)
    sc2.xy_rect(-1, 1, -1, 1, 0.0, sc2.diffuse_light((2.0, 3.0, 4.0)))
    v = _center_px(rtcheck, sc2, x=0, y=0)
    assert np.all(v == np.float32([0.25, 0.5, 0.75]))


def test_rect_axes_and_bounds(rtmi, rtcheck):
    """xy/xz/yz rects (object.cuh:105-192): each is hit only inside its bounds, from both sides."""
    for axis, lookfrom in ((0, (0, 0, 5)), (0, (0, 0, -5)), (1, (0, 5, 0.001)), (2, (5, 0, 0)), (2, (-5, 0, 0))):
        sc = _probe_scene(rtmi, bg=(0, 0, 0), lookfrom=lookfrom)
        light = sc.diffuse_light((1, 1, 1))
        (sc.xy_rect, sc.xz_rect, sc.yz_rect)[axis](-0.2, 0.2, -0.2, 0.2, 0.0, light)
        assert np.all(_center_px(rtcheck, sc, 16) == 1.0), (axis, lookfrom)
        off = np.array([rtcheck.oracle_sample(sc, 5, 0, 0, s)[0] for s in range(16)])
        assert np.all(off == 0.0), (axis, lookfrom)


def test_cylinder_closed_form(rtmi, rtcheck):
    """Open tube about z (object.cuh:233-290) rotated onto the x axis: seen from +z the
    silhouette is |y| < radius for |x| < zmax; through the open end one sees the inside."""
    sc = _probe_scene(rtmi, bg=(0, 0, 0), vfov=40.0)
    light = sc.diffuse_light((1, 2, 3))
    sc.cylinder(0.5, -1.0, 1.0, light, rotate=((0, 1, 0), 90))
    assert np.all(_center_px(rtcheck, sc, 16) == np.float32([1, 2, 3]))
    # looking down the axis from +x: open ends, the inside wall is visible off-centre only
    # (tube spans x in [-3, 1]; pixel 7 of 9 at vfov 10 has slopes 0.066..0.088 per unit
    # distance: inside the r = 0.5 opening at distance 5, through the wall before distance 9)
    sc2 = _probe_scene(rtmi, bg=(0, 0, 0), lookfrom=(6, 0, 0), vfov=10.0)
    sc2.cylinder(0.5, -3.0, 1.0, sc2.diffuse_light((1, 2, 3)), rotate=((0, 1, 0), 90))
    centre = _center_px(rtcheck, sc2, 16)
    assert np.all(centre == 0.0)  # straight through the open tube: background
    ring = np.array([rtcheck.oracle_sample(sc2, 5, 7, 4, s)[0] for s in range(16)])
    assert np.all(ring == np.float32([1, 2, 3]))  # inner wall (front_face = false) still emits


def test_translate_then_rotate_order(rtmi, rtcheck):
    """parser.hpp:423-440 applies rotate before translate whatever the key order: o2w = T*R."""
    sc = _probe_scene(rtmi, bg=(0, 0, 0), lookfrom=(2, 0, 8), lookat=(2, 0, 0), vfov=8.0)
    sc.cylinder(0.3, -0.5, 0.5, sc.diffuse_light((5, 5, 5)), rotate=((0, 1, 0), 90), translate=(2, 0, 0))
    assert np.all(_center_px(rtcheck, sc, 8) == 5.0)


def test_white_furnace_energy_conservation(rtmi, rtcheck):
    """White lambertian sphere under a uniform white background never creates energy:
    every sample is exactly 1 or 0 (depth exhaustion), never above 1."""
    sc = _probe_scene(rtmi, bg=(1, 1, 1), depth=50, vfov=30.0)
    sc.sphere((0, 0, 0), 1.0, sc.lambertian((1, 1, 1)))
    v = _center_px(rtcheck, sc, 200)
    assert v.max() <= 1.0 and v.mean() > 0.99


def test_depth_limit(rtmi, rtcheck):
    """main.cpp:20,42: max_depth bounces then black. Two facing mirrors never escape."""
    for depth in (1, 3, 50):
        sc = _probe_scene(rtmi, bg=(1, 1, 1), depth=depth)
        mirror = sc.metal((1, 1, 1), 0.0)
        sc.xy_rect(-50, 50, -50, 50, -1.0, mirror)
        sc.xy_rect(-50, 50, -50, 50, 6.0, mirror)  # behind the camera
        v, q = rtcheck.oracle_sample(sc, 1, 4, 4, 0)
        assert np.all(v == 0.0) and q == depth
    sc = _probe_scene(rtmi, bg=(1, 1, 1), depth=0)
    v, q = rtcheck.oracle_sample(sc, 1, 4, 4, 0)
    assert np.all(v == 0.0) and q == 0


def test_mirror_reflection_direction(rtmi, rtcheck):
    """metal fuzz 0 (material.h:47-53): a 45-degree bounce off a floor mirror lands on a light."""
    sc = _probe_scene(rtmi, bg=(0, 0, 0), lookfrom=(0, 2, 2), lookat=(0, 0, 0), vfov=2.0)
    sc.xz_rect(-5, 5, -5, 5, 0.0, sc.metal((0.5, 0.25, 1.0), 0.0))
    sc.xy_rect(-0.5, 0.5, 1.5, 2.5, -2.0, sc.diffuse_light((4, 4, 4)))  # mirror image of the camera
    v = _center_px(rtcheck, sc, 8)
    np.testing.assert_allclose(v, np.tile(np.float32([2, 1, 4]), (8, 1)), rtol=1e-6)


def test_dielectric_normal_incidence(rtmi, rtcheck):
    """Glass slab face-on: reflectance r0 = ((1-1.5)/(1+1.5))^2 = 0.04 decides between the
    light behind (through both faces, attenuation 1) and the black background in front."""
    sc = _probe_scene(rtmi, bg=(0, 0, 0), vfov=1.0)
    sc.sphere((0, 0, 0), 1.0, sc.dielectric(1.5))
    sc.xy_rect(-50, 50, -50, 50, -3.0, sc.diffuse_light((1, 1, 1)))
    v = _center_px(rtcheck, sc, 4000)[:, 0]
    assert set(np.unique(v)) <= {0.0, 1.0}
    # P(transmit both faces, allowing internal bounces back to the front) ~ 1 - 2*r0/(1+r0)
    assert abs(v.mean() - (1 - 2 * 0.04 / 1.04)) < 0.02


def test_checker_parity(rtmi, rtcheck):
    """checker_texture::value (texture.cuh:44-52) == sign of sin(10x)sin(10y)sin(10z)."""
    even, odd = (0.0, 1.0, 0.0), (1.0, 0.0, 0.0)
    rng = np.random.default_rng(0)
    ok = n = 0
    for _ in range(120):
        x, y = rng.uniform(-3, 3, 2)
        sc = _probe_scene(rtmi, bg=(0, 0, 0), lookfrom=(x, y, 5), lookat=(x, y, 0), vfov=0.01)
        sc.xy_rect(-50, 50, -50, 50, 0.37, sc.diffuse_light(sc.checker_texture(even, odd)))
        v = rtcheck.oracle_sample(sc, 3, 4, 4, 0)[0]
        s = math.sin(10 * x) * math.sin(10 * y) * math.sin(10 * 0.37)
        if abs(math.sin(10 * x)) < 1e-3 or abs(math.sin(10 * y)) < 1e-3:
            continue
        n += 1
        ok += int(np.all(v == np.float32(odd if s < 0 else even)))
    assert n > 100 and ok == n


def test_sky_gradient_formula(rtmi, rtcheck):
    """main.cpp:36-38: lerp(white, (0.5,0.7,1), 0.5*(unit_dir.y+1)) for an empty scene."""
    sc = _probe_scene(rtmi, sky=True, lookfrom=(0, 0, 0), lookat=(0, 1, -1), vfov=1.0)
    v = _center_px(rtcheck, sc, 4).mean(axis=0)
    t = 0.5 * (math.sqrt(0.5) + 1)
    np.testing.assert_allclose(v, [(1 - t) + t * 0.5, (1 - t) + t * 0.7, 1.0], atol=2e-3)


def test_tie_rule_later_object_wins(rtmi, rtcheck):
    """hittable_list::hit accepts root <= closest_so_far (object.cuh:28-33 with :61): two
    coincident surfaces -> the one LATER in the list is shaded, whatever its type."""
    for order in (0, 1):
        sc = _probe_scene(rtmi, bg=(0, 0, 0))
        a, b = sc.diffuse_light((1, 0, 0)), sc.diffuse_light((0, 1, 0))
        if order == 0:
            sc.xy_rect(-1, 1, -1, 1, 0.0, a)
            sc.xy_rect(-2, 2, -2, 2, 0.0, b)
            want = [0, 1, 0]
        else:
            sc.xy_rect(-2, 2, -2, 2, 0.0, b)
            sc.xy_rect(-1, 1, -1, 1, 0.0, a)
            want = [1, 0, 0]
        assert np.all(_center_px(rtcheck, sc, 4) == np.float32(want))


# ---- edge cases of the CUDA-only primitives (PARITY UNPINNED: no runnable reference; closed-form expectations) ----
def edge_case_scenes(rtmi):
    """(name, scene, expected centre-pixel radiance or None) -- also rendered on the device by
    tests/test_gpu_parity.py::test_unpinned_edge_cases_on_device."""
    out = []
    # cylinder::hit, object.cuh:262-271: the near root t0 lies beyond the tube's zmax, so the hit is re-tried with the
    # far root t1, which lies inside [zmin, zmax]: the ray enters through the open end and hits the INSIDE wall
    # (front_face = false: the normal is flipped towards the ray, the emitter still emits)
    sc = _probe_scene(rtmi, bg=(0, 0, 0), lookfrom=(0.0, 1.2, 6.0), lookat=(0.0, 0.0, 0.0), vfov=0.5)
    sc.cylinder(0.5, -4.0, 1.0, sc.diffuse_light((3, 2, 1)))  # tube about z, open end at z = 1 facing the camera
    out.append(("cylinder_far_root_inside_wall", sc, (3, 2, 1)))
    # the same tube, but the far root is outside [zmin, zmax] as well: straight through both open ends -> background
    sc = _probe_scene(rtmi, bg=(0.25, 0.5, 0.75), lookfrom=(0.0, 0.1, 6.0), lookat=(0.0, 0.0, 0.0), vfov=0.5)
    sc.cylinder(0.5, -4.0, 1.0, sc.diffuse_light((3, 2, 1)))
    out.append(("cylinder_through_both_ends", sc, (0.25, 0.5, 0.75)))
    # t == t1 after the first root fails z: `if (t == t1) return false` -- origin inside the tube wall radius, looking
    # out through the wall region beyond zmax: t0 < t_min, t = t1 is outside the z range -> miss
    sc = _probe_scene(rtmi, bg=(0.25, 0.5, 0.75), lookfrom=(0.0, 0.0, 3.0), lookat=(1.0, 0.0, 3.0), vfov=0.5)
    sc.cylinder(0.5, -1.0, 1.0, sc.diffuse_light((3, 2, 1)))
    out.append(("cylinder_from_inside_beyond_the_end", sc, (0.25, 0.5, 0.75)))
    # inside-tube normal flip with a scattering material: a mirror tube seen from inside reflects the ray back
    # through the axis onto the opposite wall... until the depth runs out: exactly max_depth queries, black
    sc = _probe_scene(rtmi, bg=(1, 1, 1), lookfrom=(0.0, 0.0, 0.0), lookat=(1.0, 0.0, 0.0), vfov=0.5, depth=7)
    sc.cylinder(0.5, -1.0, 1.0, sc.metal((1, 1, 1), 0.0))
    out.append(("mirror_tube_from_inside", sc, (0, 0, 0)))
    # rect bounds are INCLUSIVE (object.cuh:112: `x < x0 || x > x1` rejects): a ray through the exact edge x = x1 hits
    sc = _probe_scene(rtmi, bg=(0, 0, 0), lookfrom=(1.0, 0.0, 5.0), lookat=(1.0, 0.0, 0.0), vfov=1e-4)
    sc.xy_rect(-1.0, 1.0, -1.0, 1.0, 0.0, sc.diffuse_light((2, 4, 8)))
    out.append(("rect_edge_is_inside", sc, None))  # jitter moves half of the samples off the edge: checked per sample
    # a rect seen exactly edge-on (ray parallel to its plane, origin off the plane): t = +-inf -> rejected
    sc = _probe_scene(rtmi, bg=(0.25, 0.5, 0.75), lookfrom=(0.0, 0.5, 5.0), lookat=(0.0, 0.5, 0.0), vfov=1e-4)
    sc.xz_rect(-1.0, 1.0, -10.0, 10.0, 0.0, sc.diffuse_light((2, 4, 8)))
    out.append(("rect_edge_on", sc, (0.25, 0.5, 0.75)))
    return out


def test_unpinned_edge_cases(rtmi, rtcheck):
    for name, sc, want in edge_case_scenes(rtmi):
        vals = _center_px(rtcheck, sc, 32)
        if want is not None:
            assert np.all(vals == np.float32(want)), name
        elif name == "rect_edge_is_inside":
            # every sample is either the emitter (on or inside the edge) or the black background, and both occur
            hit = np.all(vals == np.float32([2, 4, 8]), axis=1)
            miss = np.all(vals == 0.0, axis=1)
            assert np.all(hit | miss) and hit.any() and miss.any(), name
    # the mirror tube really used its whole depth
    sc = [s for n, s, _ in edge_case_scenes(rtmi) if n == "mirror_tube_from_inside"][0]
    assert rtcheck.oracle_sample(sc, 5, 4, 4, 0)[1] == 7
    # exact edge, no jitter: the ray (1, 0, 5) -> (0, 0, -1) meets x = x1 exactly and is a hit (inclusive bounds)
    rect = _probe_scene(rtmi)
    rect.xy_rect(-1.0, 1.0, -1.0, 1.0, 0.0, rect.lambertian((0.5, 0.5, 0.5)))
    assert rtcheck.oracle_hit_uv(rect, (1.0, 0.0, 5.0), (0.0, 0.0, -1.0))[0]
    assert rtcheck.oracle_hit_uv(rect, (1.0, -1.0, 5.0), (0.0, 0.0, -1.0))[0]      # the corner too
    assert not rtcheck.oracle_hit_uv(rect, (np.nextafter(np.float32(1.0), np.float32(2.0)), 0.0, 5.0), (0.0, 0.0, -1.0))[0]
