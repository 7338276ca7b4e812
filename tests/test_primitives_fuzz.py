"""The restatement's ray-primitive intersections against an independent fp64 derivation, on random rays.

The rows of SURVEY 8 that no runnable reference pins (rectangles, cylinders + transforms, triangles: a6, a7, f3) are held by
closed-form known answers on chosen rays in test_primitives.py / test_textures_triangles.py.  This file adds volume: thousands
of random rays per primitive through the checker's own closest-hit probe (oracle/rt_oracle.c, rto_hit_uv), compared with the
geometry written down a second time in numpy fp64 from the reference's definitions:

  * sphere     object.cuh:47-75      nearest root of |o + t d - c|^2 = r^2 in [t_min, inf)
  * rects      object.cuh:105-192    t = (k - o_a) / d_a, the other two coordinates inside the INCLUSIVE bounds
  * cylinder   object.cuh:233-290    open tube about z in object space (o2w = T R, parser.hpp:423-440; R = Rodrigues,
                                     vec3.cuh:396-418, angle in degrees): the first of the two roots with t >= t_min and
                                     zmin <= z <= zmax; t is the same parameter in both spaces (direction not normalised)
  * triangle   hittable.py:38-71     the plane point inside the outline, from either side

An fp32 and an fp64 evaluation may disagree on hit / miss only where the fp64 geometry is within a hair of a boundary (an
edge, a grazing discriminant, t_min): those rays are set aside by a margin and counted, everything else must agree on hit /
miss and on t to 2e-5 relative (measured: 2e-6 at worst, no disagreement even among the rays set aside).  The hit record's
texture coordinates (object.cuh:87-93, 113-114, 283-288; hittable.py:54-58, 233 -- the restatement evaluates atan2 / acos with
its own fixed operation sequence, shared with the kernel) are compared on the same hits to 1e-5.  No GPU needed; the kernel is held bit-for-bit to the same restatement elsewhere."""
import numpy as np
import pytest

T_MIN = 1e-3


def _rays(rng, n, reach=4.0):
    """origins in a box around the primitive, directions towards a jittered point near it (un-normalised, any length)"""
    o = rng.uniform(-reach, reach, (n, 3))
    target = rng.normal(0.0, 0.7, (n, 3))
    d = (target - o) * rng.uniform(0.2, 3.0, (n, 1))
    return o.astype(np.float32).astype(np.float64), d.astype(np.float32).astype(np.float64)


def _probe(rtcheck, sc, o, d):
    osc = rtcheck.OracleScene(sc)
    hit, t, uv = np.zeros(len(o), bool), np.zeros(len(o)), np.zeros((len(o), 2))
    for i in range(len(o)):
        h, (u, v), tt, _ = rtcheck.oracle_hit_uv(osc, [float(v) for v in o[i]], [float(v) for v in d[i]])
        hit[i], t[i], uv[i] = h, tt, (u, v)
    _probe.uv = uv  # (texture coordinates of the last probe, for _compare_uv)
    return hit, t


def _compare(name, hit32, t32, hit64, t64, margin, min_decided=0.8):
    """margin: per ray, how far the fp64 geometry is from flipping its verdict (relative units); small ones are not judged"""
    decided = margin > 2e-3
    assert decided.mean() > min_decided, (name, decided.mean())
    bad = decided & (hit32 != hit64)
    assert not bad.any(), (name, "hit/miss differs on", int(bad.sum()), "rays, first", int(np.flatnonzero(bad)[0]))
    both = decided & hit64
    assert both.sum() > 50, (name, "too few hits to say anything", int(both.sum()))
    rel = np.abs(t32[both] - t64[both]) / np.maximum(1.0, np.abs(t64[both]))
    assert rel.max() < 2e-5, (name, rel.max())
    # and the undecided ones are few and, where both say hit, still close
    return int(both.sum()), int((~decided).sum())


def _compare_uv(name, judged, u64, v64, period=None, tol=1e-5):
    """texture coordinates of the judged hits; `period`: u wraps around (the angle's branch cut)"""
    uv = _probe.uv
    du = np.abs(uv[judged, 0] - u64[judged])
    if period:
        du = np.minimum(du, np.abs(period - du))
    dv = np.abs(uv[judged, 1] - v64[judged])
    assert du.max() < tol and dv.max() < tol, (name, du.max(), dv.max())


def _scene(rtmi):
    sc = rtmi.Scene.new(8, 8, 1, 5)
    sc.camera((0, 0, 10), (0, 0, 0), (0, 1, 0), 40.0)
    return sc, sc.lambertian(sc.solid_color((0.5, 0.5, 0.5)))


def test_sphere_random_rays(rtmi, rtcheck):
    rng = np.random.default_rng(1)
    c, r = np.array([0.3, -0.2, 0.1]), 0.9
    sc, m = _scene(rtmi)
    sc.sphere(tuple(c), r, m)
    o, d = _rays(rng, 3000)
    hit32, t32 = _probe(rtcheck, sc, o, d)
    oc = o - c
    a, hb, cc = (d * d).sum(1), (oc * d).sum(1), (oc * oc).sum(1) - r * r
    disc = hb * hb - a * cc
    sq = np.sqrt(np.maximum(disc, 0.0))
    t0, t1 = (-hb - sq) / a, (-hb + sq) / a
    t64 = np.where(t0 >= T_MIN, t0, t1)
    hit64 = (disc >= 0) & (t64 >= T_MIN)
    scale = np.maximum(hb * hb, 1e-30)
    margin = np.minimum(np.abs(disc) / scale, np.minimum(np.abs(t0 - T_MIN), np.abs(t1 - T_MIN)) / np.maximum(1.0, np.abs(t1)) * 10)
    margin = np.where(disc < 0, np.abs(disc) / scale, margin)
    _compare("sphere", hit32, t32, hit64, t64, margin)
    n = (o + t64[:, None] * d - c) / r  # outward normal, object.cuh:87-93: u = (atan2(-z, x) + pi) / 2 pi, v = acos(-y) / pi
    judged = (margin > 2e-3) & hit64
    _compare_uv("sphere", judged, (np.arctan2(-n[:, 2], n[:, 0]) + np.pi) / (2 * np.pi), np.arccos(np.clip(-n[:, 1], -1, 1)) / np.pi, period=1.0)


@pytest.mark.parametrize("axis", [0, 1, 2])
def test_rect_random_rays(rtmi, rtcheck, axis):
    rng = np.random.default_rng(10 + axis)
    a0, a1, b0, b1, k = -0.8, 1.1, -0.5, 0.9, 0.25
    sc, m = _scene(rtmi)
    # axis 0: z = k over (x, y); 1: y = k over (x, z); 2: x = k over (y, z)   (object.cuh:105-192)
    (sc.xy_rect, sc.xz_rect, sc.yz_rect)[axis](a0, a1, b0, b1, k, m)
    ka, (ia, ib) = (2, 1, 0)[axis], ((0, 1), (0, 2), (1, 2))[axis]
    o, d = _rays(rng, 3000)
    hit32, t32 = _probe(rtcheck, sc, o, d)
    with np.errstate(divide="ignore", invalid="ignore"):
        t64 = (k - o[:, ka]) / d[:, ka]
    pa, pb = o[:, ia] + t64 * d[:, ia], o[:, ib] + t64 * d[:, ib]
    inside = (pa >= a0) & (pa <= a1) & (pb >= b0) & (pb <= b1)
    hit64 = np.isfinite(t64) & (t64 >= T_MIN) & inside
    edge = np.minimum(np.minimum(np.abs(pa - a0), np.abs(pa - a1)), np.minimum(np.abs(pb - b0), np.abs(pb - b1)))
    margin = np.minimum(edge, np.abs(t64 - T_MIN) * 10)
    margin = np.where(np.isfinite(t64), margin, 1.0)
    _compare(f"rect axis {axis}", hit32, t32, hit64, np.nan_to_num(t64), margin)
    _compare_uv(f"rect axis {axis}", (margin > 2e-3) & hit64, (pa - a0) / (a1 - a0), (pb - b0) / (b1 - b0))


def _rodrigues(axis, deg):
    u = np.asarray(axis, float) / np.linalg.norm(axis)
    th = np.deg2rad(deg)
    K = np.array([[0, -u[2], u[1]], [u[2], 0, -u[0]], [-u[1], u[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)


@pytest.mark.parametrize("case", [((0, 0, 1), 0.0, (0, 0, 0)), ((1, 0, 0), 90.0, (0.2, -0.1, 0.3)), ((1, 2, 3), 37.0, (-0.4, 0.5, 0.1)),
                                  ((0, 1, 0), -120.0, (0.0, 0.3, -0.2))])
def test_cylinder_random_rays(rtmi, rtcheck, case):
    axis, deg, off = case
    rng = np.random.default_rng(int(abs(deg)) + 20)
    R_, zmin, zmax = 0.6, -0.9, 0.7
    sc, m = _scene(rtmi)
    sc.cylinder(R_, zmin, zmax, m, rotate=(axis, deg), translate=off)
    o, d = _rays(rng, 4000)
    hit32, t32 = _probe(rtcheck, sc, o, d)
    Rm = _rodrigues(axis, deg)
    oo, dd = (o - np.asarray(off)) @ Rm, d @ Rm          # R^T (o - T), R^T d  (row vectors: x @ R = R^T x)
    a = dd[:, 0] ** 2 + dd[:, 1] ** 2
    hb = oo[:, 0] * dd[:, 0] + oo[:, 1] * dd[:, 1]
    cc = oo[:, 0] ** 2 + oo[:, 1] ** 2 - R_ * R_
    disc = hb * hb - a * cc
    sq = np.sqrt(np.maximum(disc, 0.0))
    with np.errstate(divide="ignore", invalid="ignore"):
        t0, t1 = (-hb - sq) / a, (-hb + sq) / a
    z0, z1 = oo[:, 2] + t0 * dd[:, 2], oo[:, 2] + t1 * dd[:, 2]
    ok0 = (t0 >= T_MIN) & (z0 >= zmin) & (z0 <= zmax)
    ok1 = (t1 >= T_MIN) & (z1 >= zmin) & (z1 <= zmax)
    real = (disc >= 0) & (a > 1e-12)
    hit64 = real & (ok0 | ok1)
    t64 = np.nan_to_num(np.where(ok0, t0, t1))
    scale = np.maximum(hb * hb, 1e-30)

    def root_margin(t, z):
        return np.minimum(np.minimum(np.abs(z - zmin), np.abs(z - zmax)), np.abs(t - T_MIN) * 10)
    margin = np.minimum(np.abs(disc) / scale, np.minimum(root_margin(t0, z0), root_margin(t1, z1)))
    margin = np.where(disc < 0, np.abs(disc) / scale, margin)
    margin = np.where(a > 1e-9, np.nan_to_num(margin), 0.0)
    hits, undecided = _compare(f"cylinder {case}", hit32, t32, hit64, t64, margin, min_decided=0.75)
    # both kinds of hit occur: the near wall from outside and the far wall seen through the open ends / from inside
    assert (hit64 & ok0).sum() > 50 and (hit64 & ~ok0 & ok1).sum() > 20
    op = oo + t64[:, None] * dd  # object.cuh:283-288: u = (atan2(y, x) + 2 pi) / 4 pi, v = (z - zmin) / (zmax - zmin)
    _compare_uv(f"cylinder {case}", (margin > 2e-3) & hit64, (np.arctan2(op[:, 1], op[:, 0]) + 2 * np.pi) / (4 * np.pi),
                (op[:, 2] - zmin) / (zmax - zmin), period=0.5)


def test_triangle_random_rays(rtmi, rtcheck):
    rng = np.random.default_rng(40)
    v = np.array([[-0.9, -0.6, 0.1], [1.0, -0.4, -0.3], [0.1, 0.9, 0.4]])
    sc, m = _scene(rtmi)
    tuv = np.array([[0.1, 0.2], [0.9, 0.3], [0.4, 0.8]])
    sc.triangle(tuple(v[0]), tuple(v[1]), tuple(v[2]), m, u1=tuple(tuv[0]), u2=tuple(tuv[1]), u3=tuple(tuv[2]))
    o, d = _rays(rng, 4000)
    hit32, t32 = _probe(rtcheck, sc, o, d)
    e1, e2 = v[1] - v[0], v[2] - v[0]
    n = np.cross(e1, e2)
    den = d @ n
    with np.errstate(divide="ignore", invalid="ignore"):
        t64 = ((v[0] - o) @ n) / den
    p = o + t64[:, None] * d
    # barycentric coordinates of the plane point
    w = p - v[0]
    d11, d12, d22 = e1 @ e1, e1 @ e2, e2 @ e2
    w1, w2 = w @ e1, w @ e2
    det = d11 * d22 - d12 * d12
    bu, bv = (d22 * w1 - d12 * w2) / det, (d11 * w2 - d12 * w1) / det
    bw = 1.0 - bu - bv
    inside = (bu > 0) & (bv > 0) & (bw > 0)
    hit64 = np.isfinite(t64) & (t64 >= T_MIN) & inside
    margin = np.minimum(np.minimum(np.abs(bu), np.abs(bv)), np.abs(bw))
    margin = np.minimum(margin, np.abs(t64 - T_MIN) * 10)
    margin = np.minimum(margin, np.abs(den) / (np.linalg.norm(d, axis=1) * np.linalg.norm(n)) * 10)   # grazing the plane
    margin = np.nan_to_num(margin)
    hits, _ = _compare("triangle", hit32, t32, hit64, np.nan_to_num(t64), margin, min_decided=0.9)
    # seen from both sides
    assert ((den > 0) & hit64).sum() > 30 and ((den < 0) & hit64).sum() > 30
    # hittable.py:54-58, 233: the area weights w1 = |a1 x a2| (the sub-triangle opposite v3), w2 = |a1 x a3| (opposite v2),
    # w3 = |a3 x a2| (opposite v1) multiply u1, u2, u3 IN THAT ORDER -- restated as the reference has it
    l1, l2, l3 = bw, bu, bv  # barycentric weights of v1, v2, v3
    _compare_uv("triangle", (margin > 2e-3) & hit64, tuv[0, 0] * l3 + tuv[1, 0] * l2 + tuv[2, 0] * l1,
                tuv[0, 1] * l3 + tuv[1, 1] * l2 + tuv[2, 1] * l1)


# ---- the hit record's point and normal, seen through the scatter step -------------------------------------------------------
# lambertian::scatter (material.h:25-35) sends the next ray from the hit point p along  n + unit_vector  with n the unit normal
# turned against the incoming ray (set_face_normal, hittable.h:17-20): whatever the random unit vector is, the next query starts
# at p and |dir - n| = 1.  So the checker's trace of a path (rto_trace_sample: origin and direction of every query) pins p, the
# direction of n, its sign and its length against the fp64 geometry -- for the primitives whose hit() no reference run covers.

def _normal_check(rtmi, rtcheck, build, normal_at, lookfroms, name, min_hits=300):
    total = 0
    for cam in lookfroms:
        sc = rtmi.Scene.new(40, 40, 1, 4)
        sc.set_background((0.7, 0.8, 1.0), sky_gradient=True, defocus_blur=False)
        sc.camera(cam, (0.05, 0.02, -0.03), (0, 1, 0), 38.0)
        build(sc, sc.lambertian(sc.solid_color((0.6, 0.5, 0.4))))
        osc = rtcheck.OracleScene(sc)
        for y in range(40):
            for x in range(40):
                _, q = rtcheck.oracle_trace_sample(osc, 5, x, y, 0, max_queries=3)
                if len(q) < 2 or q[0, 7] == 0.0:
                    continue
                o, d, t = q[0, 0:3].astype(np.float64), q[0, 3:6].astype(np.float64), float(q[0, 6])
                p = o + t * d
                assert np.abs(q[1, 0:3] - p).max() < 1e-5 * max(1.0, np.abs(p).max()), (name, "hit point", x, y)
                n = np.asarray(normal_at(p), float)
                n = n / np.linalg.norm(n)
                if np.dot(n, d) > 0:
                    n = -n  # against the incoming ray
                dev = abs(np.linalg.norm(q[1, 3:6].astype(np.float64) - n) - 1.0)
                assert dev < 2e-5, (name, "normal", x, y, dev)
                total += 1
    assert total >= min_hits, (name, total)


@pytest.mark.parametrize("axis", [0, 1, 2])
def test_rect_normals(rtmi, rtcheck, axis):
    nrm = ((0, 0, 1), (0, 1, 0), (1, 0, 0))[axis]  # object.cuh:117, 154, 187: the +axis normal, then set_face_normal
    build = lambda sc, m: (sc.xy_rect, sc.xz_rect, sc.yz_rect)[axis](-0.9, 1.0, -0.7, 0.8, 0.1, m)
    cams = [tuple(4.0 * np.array(v)) for v in ((0.6, 0.5, 0.7), (-0.5, -0.6, -0.7), (0.7, -0.4, 0.5))]
    _normal_check(rtmi, rtcheck, build, lambda p: nrm, cams, f"rect axis {axis}")


@pytest.mark.parametrize("case", [((0, 0, 1), 0.0, (0, 0, 0)), ((1, 2, 3), 37.0, (-0.2, 0.1, 0.1)), ((1, 0, 0), 90.0, (0.1, -0.1, 0.2))])
def test_cylinder_normals(rtmi, rtcheck, case):
    axis, deg, off = case
    Rm = _rodrigues(axis, deg)
    build = lambda sc, m: sc.cylinder(0.6, -0.8, 0.7, m, rotate=(axis, deg), translate=off)

    def normal_at(p):  # object.cuh:276-281: (x, y, 0) in object space, carried to world space by the rotation
        q = (np.asarray(p) - np.asarray(off)) @ Rm
        return Rm @ np.array([q[0], q[1], 0.0])
    cams = [(2.5, 1.5, 3.0), (-3.0, -1.0, 2.0), (0.3, 3.5, 0.8), (0.2, -0.4, 4.0)]  # outside views and down the open tube
    _normal_check(rtmi, rtcheck, build, normal_at, cams, f"cylinder {case}")


def test_triangle_normals(rtmi, rtcheck):
    v = np.array([[-0.9, -0.6, 0.1], [1.0, -0.4, -0.3], [0.1, 0.9, 0.4]])
    n = np.cross(v[1] - v[0], v[2] - v[0])
    build = lambda sc, m: sc.triangle(tuple(v[0]), tuple(v[1]), tuple(v[2]), m)
    _normal_check(rtmi, rtcheck, build, lambda p: n, [(1.0, 0.8, 3.0), (-0.5, -1.0, -3.0), (2.0, 0.2, 1.5)], "triangle", min_hits=200)


def test_sphere_normals_inside_and_out(rtmi, rtcheck):
    c, r = np.array([0.1, -0.1, 0.0]), 0.8
    build = lambda sc, m: sc.sphere(tuple(c), r, m)
    _normal_check(rtmi, rtcheck, build, lambda p: np.asarray(p) - c, [(2.0, 1.0, 2.5), (0.15, -0.05, 0.1)], "sphere")
