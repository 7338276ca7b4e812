"""-m gpu: the uniform grid lists EVERY primitive type (taichi-version/bvh.py:109-199 indexes every hittable;
gpu-version/main.cu:426 wraps the whole hittable_list).  Rectangles, cylinders and triangles that are not oversized are
entries of the cells of the wide grid tables, tested by the lanes whose rays cross those cells; the kernel must equal the
reference's linear scan (rt_opts.variant 16) on the whole frame and the CPU checker (oracle/) on sampled rows, bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED = 2023


def _rows_equal_checker(rtcheck, sc, img, seed, rows):
    osc = rtcheck.OracleScene(sc)
    for y in rows:
        ref, _ = rtcheck.oracle_render(osc, seed=seed, rows=(y, y + 1))
        assert np.array_equal(img[y], ref[y]), f"row {y} differs from the CPU checker"


def height_field(rtmi, n, w, h, spp, depth=8, spheres=200, seed=4, extent=10.0):
    """n x n quads = 2 n^2 triangles over [-extent/2, extent/2]^2, a ground sphere, and `spheres` small spheres above."""
    sc = rtmi.Scene.new(w, h, spp, depth)
    sc.set_background((0.7, 0.8, 1.0), sky_gradient=True, defocus_blur=False)
    sc.camera((0.9 * extent, 0.55 * extent, 1.1 * extent), (0, 0.3, 0), (0, 1, 0), 35.0)
    rng = np.random.default_rng(seed)
    mats = [sc.lambertian((0.7, 0.3, 0.3)), sc.metal((0.8, 0.8, 0.8), 0.05), sc.lambertian((0.3, 0.6, 0.3)), sc.dielectric(1.5)]
    step = extent / n
    hgt = rng.uniform(0.0, 2.5 * step, size=(n + 1, n + 1)) + 0.4 * np.sin(np.arange(n + 1) * 6.0 / n)[:, None]
    P = lambda i, j: (float(i * step - extent / 2), float(hgt[i, j]), float(j * step - extent / 2))
    for i in range(n):
        for j in range(n):
            m = mats[(i + 2 * j) % 3]
            sc.triangle(P(i, j), P(i + 1, j), P(i, j + 1), m)
            sc.triangle(P(i + 1, j), P(i + 1, j + 1), P(i, j + 1), m)
    sc.sphere((0, -1000.5, 0), 1000.0, mats[2])
    for k in range(spheres):
        c = rng.uniform(-extent / 2, extent / 2, 3)
        sc.sphere((float(c[0]), float(1.2 + 0.8 * rng.random()), float(c[2])), float(rng.uniform(0.05, 0.15)), mats[k % 4])
    return sc


def test_twenty_thousand_triangles_through_the_grid(rtmi, rtcheck):
    """a height field of 20 000 triangles plus spheres at 640 x 360 x 4 -- the grid walk equals the
    linear scan on the whole frame and the CPU checker on sampled rows."""
    sc = height_field(rtmi, 100, 640, 360, 4)
    assert sc.info.num_prims == 20000 + 1 + 200
    st = rtmi.Stats()
    img = sc.render(rtmi.Opts(seed=SEED), st)
    assert st.kernel_variant == 44                              # tables too large for LDS: global memory
    flat = sc.render(rtmi.Opts(seed=SEED, variant=24))           # the linear scan, tables in global memory at this size
    assert np.array_equal(img, flat), f"{(img != flat).any(axis=2).sum()} pixels differ from the linear scan"
    _rows_equal_checker(rtcheck, sc, img, SEED, (7, 180, 301))
    c = sc.count(rtmi.Opts(seed=SEED, tile_rows=8, tile_first=20, tile_stride=100000))
    assert c.cull_mode == 7 and c.kernel_variant == 44
    # the walk tests a handful of primitives per query, not twenty thousand
    assert 0 < c.lane_clusters < 40 * c.queries


def test_small_mesh_and_mixed_scene_in_lds(rtmi, rtcheck):
    """2 x 6 x 6 triangles, 15 thin cylinders, 12 small rectangles and 60 spheres inside a room of two big walls: the
    wide tables fit LDS (variant 36); the walls are oversized and tested for every query; same image from global-memory
    tables (44), the linear scan (16) and the checker (whole frame)."""
    sc = height_field(rtmi, 6, 96, 60, 3, spheres=60, extent=6.0)
    rng = np.random.default_rng(8)
    light = sc.diffuse_light((3.0, 2.5, 2.0))
    glass = sc.dielectric(1.5)
    red = sc.lambertian((0.8, 0.2, 0.2))
    for k in range(15):
        c = rng.uniform(-2.5, 2.5, 3)
        axis = rng.normal(size=3)
        sc.cylinder(float(rng.uniform(0.03, 0.1)), -0.4, 0.4, [light, glass, red][k % 3],
                    rotate=(tuple(axis / np.linalg.norm(axis)), float(rng.uniform(0, 180))),
                    translate=(float(c[0]), float(1.5 + 0.5 * c[1] / 2.5), float(c[2])))
    for k in range(12):
        c = rng.uniform(-2.5, 2.5, 3)
        s = float(rng.uniform(0.1, 0.3))
        [sc.xy_rect, sc.xz_rect, sc.yz_rect][k % 3](float(c[0]), float(c[0] + s), float(c[1]), float(c[1] + s), float(2.0 + 0.2 * c[2]), red if k % 2 else light)
    mirror = sc.metal((0.9, 0.9, 0.9), 0.0)
    sc.xy_rect(-40.0, 40.0, -1.0, 40.0, -6.0, mirror)   # two walls: oversized
    sc.yz_rect(-1.0, 40.0, -40.0, 40.0, -6.0, red)
    st = rtmi.Stats()
    img = sc.render(rtmi.Opts(seed=SEED), st)
    assert st.kernel_variant == 36
    ref, _ = rtcheck.oracle_render(sc, seed=SEED)
    assert np.array_equal(img, ref), f"{(img != ref).any(axis=2).sum()} pixels differ from the CPU checker"
    for v in (44, 16, 24):  # (the cluster searches have no build with triangles)
        assert np.array_equal(img, sc.render(rtmi.Opts(seed=SEED, variant=v))), v
    c, cimg = sc.count(rtmi.Opts(seed=SEED), want_image=True)
    assert np.array_equal(cimg, img) and c.kernel_variant == 36 and c.cull_mode == 7
    _, want = rtcheck.oracle_render(sc, seed=SEED, want_counts=True)
    got = c.as_dict()
    for k, v in want.items():
        assert got[k] == v, k


@pytest.mark.parametrize("fuzz_seed", [31, 32, 33])
def test_random_mixed_scenes_equal_the_flat_scan(rtmi, rtcheck, fuzz_seed):
    """Random scenes of every primitive type -- sheets and volumes, 30 to 1500 primitives, cameras inside and outside,
    mirrors that send rays back from far away (the far tier of the lists, the scan from beyond their reach) -- the grid
    walk (LDS and global tables) against the flat scan on the whole frame and the checker on three rows."""
    rng = np.random.default_rng(fuzz_seed)
    for case in range(6):
        n = int(rng.choice([30, 120, 500, 1500]))
        sheet = rng.random() < 0.4
        half = float(rng.uniform(2.0, 10.0))
        w, h, spp = int(rng.integers(40, 100)), int(rng.integers(24, 56)), int(rng.integers(1, 4))
        sc = rtmi.Scene.new(w, h, spp, int(rng.integers(2, 10)))
        inside = rng.random() < 0.3
        eye = rng.uniform(-0.5, 0.5, 3) * half if inside else rng.uniform(1.5, 3.0) * half * np.array([rng.choice([-1, 1]), 0.4, rng.choice([-1, 1])])
        sc.camera(tuple(eye), tuple(rng.uniform(-0.2, 0.2, 3) * half), (0, 1, 0), float(rng.uniform(25, 70)), 0.0,
                  float(rng.choice([0.0, 0.1])), 0.0)
        sc.set_background(tuple(rng.uniform(0.3, 1.0, 3)), sky_gradient=bool(rng.random() < 0.5), defocus_blur=bool(rng.random() < 0.5))
        mats = [sc.lambertian(tuple(rng.uniform(0.1, 0.9, 3))) for _ in range(3)]
        mats += [sc.metal(tuple(rng.uniform(0.5, 1.0, 3)), float(rng.choice([0.0, 0.2]))), sc.dielectric(1.5),
                 sc.diffuse_light(tuple(rng.uniform(0.5, 3.0, 3))), sc.lambertian(sc.checker_texture((0.2, 0.3, 0.1), (0.9, 0.9, 0.9)))]
        if rng.random() < 0.6:
            sc.sphere((0.0, -1000.0 - (0.0 if sheet else half), 0.0), 1000.0, mats[-1])
        if rng.random() < 0.5:  # a far mirror wall: origins in the far tier or beyond the lists' reach
            sc.xy_rect(-500.0, 500.0, -500.0, 500.0, float(-rng.choice([5.0, 12.0, 30.0]) * half), sc.metal((0.95, 0.95, 0.95), 0.0))
        size = float(rng.uniform(0.05, 0.3)) * (1.0 if n < 600 else 0.5)
        kinds = rng.choice(4, size=n, p=[0.35, 0.15, 0.2, 0.3])
        for i in range(n):
            c = rng.uniform(-half, half, 3)
            s = float(rng.uniform(0.3 * size, size))
            if sheet:
                c[1] = s
            m = mats[i % len(mats)]
            if kinds[i] == 0:
                sc.sphere(tuple(c), s, m)
            elif kinds[i] == 1:
                [sc.xy_rect, sc.xz_rect, sc.yz_rect][i % 3](float(c[0]), float(c[0] + 2 * s), float(c[1]), float(c[1] + 2 * s), float(c[2]), m)
            elif kinds[i] == 2:
                axis = rng.normal(size=3)
                sc.cylinder(0.4 * s, -s, s, m, rotate=(tuple(axis / np.linalg.norm(axis)), float(rng.uniform(0, 180))),
                            translate=tuple(float(v) for v in c))
            else:
                a, b = rng.normal(size=3) * s, rng.normal(size=3) * s
                sc.triangle(tuple(c), tuple(c + a), tuple(c + b), m)
        st = rtmi.Stats()
        img = sc.render(rtmi.Opts(seed=case), st)
        flat = sc.render(rtmi.Opts(seed=case, variant=24))
        what = f"case {case}: n {n} sheet {sheet} half {half:.1f} inside {inside} {w}x{h}x{spp} kernel {st.kernel_variant}"
        assert np.array_equal(img, flat), f"{what}: {(img != flat).any(axis=2).sum()} pixels differ from the flat scan"
        for v in (36, 44):
            try:
                other = sc.render(rtmi.Opts(seed=case, variant=v))
            except rtmi.RtmiError as e:
                assert e.status == 6, what
                continue
            assert np.array_equal(other, flat), f"{what}: variant {v}"
        _rows_equal_checker(rtcheck, sc, flat, case, sorted({0, h // 3, h - 1}))


def test_counting_kernel_reports_the_kernel_that_ran(rtmi, rtcheck):
    """rt_render_hip_count launches a counting build of the grid walks or of the cluster searches 64 / 128; for
    any other requested variant it counts with the kernel variant 0 would run -- sized for THAT kernel -- and says so."""
    sc = rtmi.Scene.rtiow(7, 96, 54, 3, 20)
    img = sc.render(rtmi.Opts(seed=SEED, variant=16))
    for asked, ran, mode in ((16, 6, 5), (0, 6, 5), (2, 6, 5), (17, 6, 5), (32, 6, 5), (64, 64, 2), (128, 128, 3)):
        st, cimg = sc.count(rtmi.Opts(seed=SEED, variant=asked), want_image=True)
        assert st.kernel_variant == ran and st.cull_mode == mode, asked
        assert np.array_equal(cimg, img), asked
    mixed = rtmi.Scene.dna(30.0)
    mixed.override(width=64, height=36, spp=2)
    ref = mixed.render(rtmi.Opts(seed=SEED, variant=16))
    for asked in (0, 16, 24, 44):
        st, cimg = mixed.count(rtmi.Opts(seed=SEED, variant=asked), want_image=True)
        assert st.kernel_variant == (44 if asked == 44 else 36) and st.cull_mode == 7, asked
        assert np.array_equal(cimg, ref), asked


def test_grid_of_other_primitives_only_and_rays_along_its_planes(rtmi, rtcheck):
    """No sphere at all: the grid is built over rectangles, triangles and cylinders alone.  The primitives sit ON the lattice
    planes of a regular arrangement (so their boxes end exactly on cell boundaries), the camera's rays have direction
    components of exactly 0 (they never leave their cell column), a long thin cylinder crosses the whole grid and a
    triangle far larger than the rest is tested per query."""
    sc = rtmi.Scene.new(9, 9, 4, 6)
    sc.set_background((0.25, 0.5, 0.75), sky_gradient=False, defocus_blur=False)
    sc.camera((1.0, 1.0, 30.0), (1.0, 1.0, 0.0), (0, 1, 0), 1e-6, 1.0, 0.0, 30.0)  # every ray: d = (0, 0, -1) exactly
    lights = [sc.diffuse_light((1.0 + k, 2.0, 3.0)) for k in range(3)]
    mirror = sc.metal((1, 1, 1), 0.0)
    k = 0
    for ix in range(-3, 5):
        for iy in range(-3, 5):
            for iz in range(3):
                x, y, z = float(ix), float(iy), float(-2 * iz)
                m = lights[k % 3] if (ix, iy) != (1, 1) or iz else mirror  # the camera's column: a mirror first
                if k % 3 == 0:
                    sc.xy_rect(x - 0.5, x + 0.5, y - 0.5, y + 0.5, z, m)
                elif k % 3 == 1:
                    sc.triangle((x - 0.5, y - 0.5, z), (x + 0.5, y - 0.5, z), (x, y + 0.5, z), m)
                else:
                    sc.cylinder(0.4, -0.5, 0.5, m, rotate=((1.0, 0.0, 0.0), 90.0), translate=(x, y, z))
                k += 1
    sc.cylinder(0.05, -40.0, 40.0, lights[0], rotate=((0.0, 1.0, 0.0), 90.0), translate=(0.0, 2.5, -1.0))  # through every column
    sc.triangle((-60.0, -60.0, -9.0), (60.0, -60.0, -9.0), (0.0, 80.0, -9.0), lights[2])                   # a backdrop: oversized
    assert all(int(t) != 0 for t in sc.prims()["type"])
    st = rtmi.Stats()
    img = sc.render(rtmi.Opts(seed=SEED), st)
    assert st.kernel_variant == 44                      # (192 primitives of 5 - 6 records each: beyond the LDS budget)
    ref, _ = rtcheck.oracle_render(sc, seed=SEED)
    assert np.array_equal(img, ref)
    assert np.array_equal(img, sc.render(rtmi.Opts(seed=SEED, variant=16)))
    assert np.array_equal(img, sc.render(rtmi.Opts(seed=SEED, variant=36)))  # the same tables from LDS
    c = sc.count(rtmi.Opts(seed=SEED))
    assert c.cull_mode == 7 and c.lane_groups > 0 and c.lane_clusters > 0   # lanes entered the grid and tested its entries
    # a wider view of the same arrangement, from inside it
    sc.camera((0.3, 0.2, -1.0), (3.0, 2.0, -3.0), (0, 1, 0), 80.0)
    sc.override(width=64, height=48, spp=3)
    img = sc.render(rtmi.Opts(seed=SEED))
    ref, _ = rtcheck.oracle_render(sc, seed=SEED)
    assert np.array_equal(img, ref)
