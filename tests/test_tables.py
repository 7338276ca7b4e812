"""Host logic of the candidate search (no GPU): the tables pack_scene (csrc/render_host.hip) builds, read back through
rt_scene_table_info / rt_scene_table_image and checked against the geometry.  The kernel only finds what a cell lists, so the
invariant that matters is COVERAGE: every cell a primitive's exact box touches lists it (the lists are built from grown boxes:
a superset), in the tier a ray origin near the cloud reads -- or the primitive is in the always-tested part of its table."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Tables:
    def __init__(self, sc):
        self.sc = sc
        self.t = t = sc.table_info()
        self.img = img = sc.table_image()
        assert img.shape == (t.image_floats // 4, 4)
        self.u32 = img.view(np.uint32).reshape(-1)
        self.n = np.array(list(t.grid_n), dtype=np.int64)
        self.gmin = np.array(list(t.grid_min), dtype=np.float64)
        self.gsize = np.array(list(t.grid_size), dtype=np.float64)
        ncell = t.grid_cells
        assert ncell == (int(self.n.prod()) if ncell else 0)
        if t.grid_wide:
            words = self.u32[t.off_grid_cells * 4: t.off_grid_cells * 4 + 2 * ncell].reshape(-1, 2)
            self.first = words[:, 0].astype(np.int64)
            self.n_near = (words[:, 1] & 1023).astype(np.int64)
            self.n_all = ((words[:, 1] >> 10) & 1023).astype(np.int64)
            self.n_other = (words[:, 1] >> 20).astype(np.int64)
            self.items = self.u32[t.off_grid_items * 4:].astype(np.int64)
        else:
            words = self.u32[t.off_grid_cells * 4: t.off_grid_cells * 4 + ncell]
            self.first = (words >> 12).astype(np.int64)
            self.n_near = ((words >> 6) & 63).astype(np.int64)
            self.n_all = (words & 63).astype(np.int64)
            self.n_other = np.zeros(ncell, dtype=np.int64)
            self.items = img.view(np.uint16).reshape(-1)[t.off_grid_items * 8:].astype(np.int64)
        self.prims = sc.prims()

    def cells_of_box(self, lo, hi):
        """indices of the cells an axis-aligned box touches (the packer's cell_of, clamped to the grid)"""
        i0 = np.clip(np.floor((np.asarray(lo, dtype=np.float64) - self.gmin) / self.gsize), 0, self.n - 1).astype(np.int64)
        i1 = np.clip(np.floor((np.asarray(hi, dtype=np.float64) - self.gmin) / self.gsize), 0, self.n - 1).astype(np.int64)
        nx, ny = int(self.n[0]), int(self.n[1])
        return [(iz * ny + iy) * nx + ix for iz in range(i0[2], i1[2] + 1) for iy in range(i0[1], i1[1] + 1)
                for ix in range(i0[0], i1[0] + 1)]

    def near(self, c):
        return self.items[self.first[c]: self.first[c] + self.n_near[c]]

    def all_spheres(self, c):
        return self.items[self.first[c]: self.first[c] + self.n_all[c]]

    def others(self, c):
        s = self.first[c] + self.n_all[c]
        return self.items[s: s + self.n_other[c]]


def other_box(p):
    """exact world box of a rectangle / triangle; a cylinder's from points on its surface (a lower bound: coverage test)"""
    typ = int(p["type"])
    f, m = p["f"].astype(np.float64), p["m"].astype(np.float64)
    if typ in (1, 2, 3):
        ia, ib, ik = {1: (0, 1, 2), 2: (0, 2, 1), 3: (1, 2, 0)}[typ]
        lo, hi = np.zeros(3), np.zeros(3)
        lo[ia], hi[ia] = min(f[0], f[1]), max(f[0], f[1])
        lo[ib], hi[ib] = min(f[2], f[3]), max(f[2], f[3])
        lo[ik] = hi[ik] = f[4]
        return lo, hi
    if typ == 5:
        v = m[:9].reshape(3, 3)
        return v.min(axis=0), v.max(axis=0)
    ang = np.linspace(0, 2 * np.pi, 64, endpoint=False)
    pts = np.array([[abs(f[0]) * np.cos(a), abs(f[0]) * np.sin(a), z, 1.0] for a in ang for z in (f[1], f[2])])
    w = pts @ m.reshape(3, 4).T
    return w.min(axis=0), w.max(axis=0)


def check_coverage(sc, expect_variant=None):
    T = Tables(sc)
    t, img, P = T.t, T.img, T.prims
    if expect_variant is not None:
        assert t.kernel_variant == expect_variant
    ns = t.ns
    # ---- sphere slots: every sphere once; padding slots never hit
    slot_prim = {}
    for slot in range(ns):
        if np.isneginf(img[slot, 3]):
            continue
        prim = int(img[t.off_sph_cold + slot].view(np.uint32)[2])
        assert int(P["type"][prim]) == 0 and prim not in slot_prim.values()
        slot_prim[slot] = prim
        assert np.array_equal(img[slot, :3], P["f"][prim][:3]) and img[slot, 3] == np.float32(P["f"][prim][3]) * np.float32(P["f"][prim][3])
    assert sorted(slot_prim.values()) == [i for i in range(len(P)) if int(P["type"][i]) == 0]
    # ---- other primitives: grouped id -> list index
    gid_prim = {}
    for j in range(t.nr):
        gid_prim[ns + j] = int(img[t.off_rect_cold + j].view(np.uint32)[1])
    for k in range(t.nc):
        gid_prim[ns + t.nr + k] = int(img[t.off_cyl_cold + 4 * k + 3].view(np.uint32)[1])
    for k in range(t.nt):
        gid_prim[ns + t.nr + t.nc + k] = int(img[t.off_tri_cold + 2 * k].view(np.uint32)[1])
    assert sorted(gid_prim.values()) == [i for i in range(len(P)) if int(P["type"][i]) != 0]
    always = set(range(ns, ns + t.nr_a)) | set(range(ns + t.nr, ns + t.nr + t.nc_a)) | set(range(ns + t.nr + t.nc, ns + t.nr + t.nc + t.nt_a))
    if not t.grid_wide:
        assert t.nr + t.nc + t.nt == 0
    # ---- the lists only name what can be listed
    for c in range(t.grid_cells):
        sp = T.all_spheres(c)
        assert len(set(sp)) == len(sp) and all(t.np <= s < ns and s in slot_prim for s in sp)
        ot = T.others(c)
        assert len(set(ot)) == len(ot) and all(g in gid_prim and g not in always for g in ot)
    if t.grid_cells == 0:
        assert len(slot_prim) == sum(1 for s in slot_prim if s < t.np) and len(always) == len(gid_prim)  # everything is tested per query
        return T
    # ---- coverage: the cells an exact box touches list the primitive (spheres: already in the near tier)
    for slot, prim in slot_prim.items():
        if slot < t.np:
            continue
        c3, r = P["f"][prim][:3].astype(np.float64), abs(float(P["f"][prim][3]))
        for c in T.cells_of_box(c3 - r, c3 + r):
            assert slot in T.near(c), f"sphere {prim} (slot {slot}) missing from cell {c}"
    for g, prim in gid_prim.items():
        if g in always:
            continue
        lo, hi = other_box(P[prim])
        for c in T.cells_of_box(lo, hi):
            assert g in T.others(c), f"primitive {prim} (id {g}) missing from cell {c}"
        # its stored box (behind its records) holds it
        if g >= ns + t.nr:
            k = g - ns - t.nr
            base = t.off_cyl_hot + 6 * k + 4 if k < t.nc else t.off_tri_hot + 5 * (k - t.nc) + 3
            assert np.all(img[base, :3] <= lo + 1e-6) and np.all(img[base + 1, :3] >= hi - 1e-6)
    return T


def _mixed(rtmi, n, seed, half=5.0, sheet=False, size=0.25):
    rng = np.random.default_rng(seed)
    sc = rtmi.Scene.new(32, 20, 1, 5)
    sc.camera((0, 2, 3 * half), (0, 0, 0), (0, 1, 0), 40.0)
    m = sc.lambertian((0.5, 0.5, 0.5))
    kinds = rng.choice(4, size=n, p=[0.4, 0.15, 0.2, 0.25])
    for i in range(n):
        c = rng.uniform(-half, half, 3)
        s = float(rng.uniform(0.3 * size, size))
        if sheet:
            c[1] = s
        if kinds[i] == 0:
            sc.sphere(tuple(c), s, m)
        elif kinds[i] == 1:
            [sc.xy_rect, sc.xz_rect, sc.yz_rect][i % 3](float(c[0]), float(c[0] + 2 * s), float(c[1]), float(c[1] + 2 * s), float(c[2]), m)
        elif kinds[i] == 2:
            ax = rng.normal(size=3)
            sc.cylinder(0.4 * s, -s, s, m, rotate=(tuple(ax / np.linalg.norm(ax)), float(rng.uniform(0, 180))), translate=tuple(float(v) for v in c))
        else:
            a, b = rng.normal(size=3) * s, rng.normal(size=3) * s
            sc.triangle(tuple(c), tuple(c + a), tuple(c + b), m)
    return sc, m


def test_rtiow_tables(rtmi):
    """The headline scene: compact tables, a grid one cell high, the ground and the three big spheres tested per query."""
    T = check_coverage(rtmi.Scene.rtiow(7, 1920, 1080, 1024, 50), expect_variant=2)
    t = T.t
    assert (t.grid_wide, t.grid_sheet, t.np, t.ncl) == (0, 1, 4, 60) and list(t.grid_n) == [18, 1, 18]
    assert t.hot_bytes_grid <= 17261 and T.n_all.max() <= 63 and int(T.n_all.sum()) == 956  # three list entries per cell


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_mixed_scene_tables_cover_every_primitive(rtmi, seed):
    for n, sheet in ((40, False), (300, True), (1200, False)):
        sc, m = _mixed(rtmi, n, seed * 10 + n, sheet=sheet, size=0.25 if n < 1000 else 0.12)
        sc.xy_rect(-200, 200, -200, 200, -30.0, m)                       # a wall: oversized, tested per query
        sc.sphere((0, -1000 - 5.0, 0), 1000.0, m)                        # the ground: always-tested sphere
        T = check_coverage(sc)
        t = T.t
        assert t.grid_wide == 1 and t.grid_cells > 0 and t.nr_a >= 1 and t.np >= 1
        assert t.kernel_variant == (36 if t.hot_bytes_grid <= 17261 else 44)


def test_small_and_sphere_only_scenes(rtmi, scenes_dir):
    sc = rtmi.Scene.load(os.path.join(scenes_dir, "three_sphere.json"))
    T = check_coverage(sc, expect_variant=6)                             # five spheres: nothing listed, compact format
    assert T.t.grid_cells == 0 and T.t.np >= 5
    T = check_coverage(rtmi.Scene.load(os.path.join(scenes_dir, "mixed_emissive.json")), expect_variant=16)
    assert T.t.grid_cells == 0 and T.t.grid_wide == 1                    # a handful of primitives of several types: the scan
    T = check_coverage(rtmi.Scene.dna(30.0), expect_variant=36)          # 60 spheres + 30 cylinders: cylinders in the cells
    assert T.t.nc == 30 and T.t.nc_a == 0 and int(T.n_other.sum()) >= 30
    vol = rtmi.Scene.new(16, 16, 1)
    vol.camera((0, 0, 12), (0, 0, 0), (0, 1, 0), 40.0)
    m = vol.lambertian((0.5, 0.5, 0.5))
    rng = np.random.default_rng(5)
    for _ in range(300):
        vol.sphere(tuple(rng.uniform(-3, 3, 3)), 0.15, m)
    T = check_coverage(vol, expect_variant=6)                            # a sphere volume that fits LDS: compact, 3-D
    assert T.t.grid_sheet == 0 and T.t.grid_n[1] > 1
    for _ in range(3000):
        vol.sphere(tuple(rng.uniform(-3, 3, 3)), 0.05, m)
    T = check_coverage(vol, expect_variant=44)                           # beyond LDS: wide tables from global memory
    assert T.t.grid_wide == 1 and T.t.hot_bytes_grid > 17261


def test_clumps(rtmi):
    """More than 63 spheres through one cell: wide tables.  More than 1023: the clump is tested per query and the rest keeps
    its grid; the packer terminates on a scene that is one clump."""
    rng = np.random.default_rng(7)
    sc, m = _mixed(rtmi, 0, 0)
    for _ in range(200):
        sc.sphere(tuple(rng.uniform(-4, 4, 3)), 0.1, m)
    for _ in range(90):
        sc.sphere(tuple(np.array([1.0, 1.0, 1.0]) + rng.uniform(-0.02, 0.02, 3)), 0.1, m)
    T = check_coverage(sc, expect_variant=36)
    assert T.t.grid_wide == 1 and T.n_all.max() >= 90
    for _ in range(1100):
        sc.sphere(tuple(np.array([-1.0, 0.5, 1.0]) + rng.uniform(-0.01, 0.01, 3)), 0.05, m)
    T = check_coverage(sc)
    assert T.t.np >= 1100 and T.t.grid_cells > 0 and T.n_all.max() <= 1023
    one = rtmi.Scene.new(16, 16, 1)
    one.camera((0, 0, 5), (0, 0, 0), (0, 1, 0), 40.0)
    m = one.lambertian((0.5, 0.5, 0.5))
    for _ in range(1500):
        one.sphere((0.0, 0.0, 0.0), 0.5, m)                              # 1500 copies of one sphere
    T = check_coverage(one)
    assert T.t.np >= 1500 and T.t.ncl == 0                               # the reference's scan, as the limit case


def test_mesh_tables(rtmi):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_grid_all import height_field
    T = check_coverage(height_field(rtmi, 24, 64, 36, 1, spheres=50), expect_variant=44)
    t = T.t
    assert t.nt == 2 * 24 * 24 and t.nt_a == 0 and t.grid_wide == 1
    per_cell = T.n_other[T.n_other > 0]
    assert 1 <= per_cell.mean() <= 16                                    # a handful of triangles per occupied cell
