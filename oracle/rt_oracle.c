/*
 * rt_oracle.c -- TEST INFRASTRUCTURE ONLY (see rt_oracle.h).
 *
 * A CPU restatement, in scalar fp32 C, of the reference's per-pixel path tracing
 * hot path.  It keeps the reference's STRUCTURE (array-of-structs objects visited
 * in list order, every object's hit() filling a full hit_record that the list
 * copies on improvement, material scatter functions returning attenuation +
 * scattered ray) so that each function can be read against the reference lines
 * it cites.  Paths are relative to the reference checkout:
 *   cpu/  = cmake-cpu-version/   (fp64, spheres only; the parity oracle)
 *   gpu/  = gpu-version/         (fp32 CUDA; rects, cylinders, emission, JSON)
 *
 * What is restated, not copied: the arithmetic is fp32 with an explicit
 * operation order (every fused multiply-add is written as fmaf, nothing else may
 * contract: build with -ffp-contract=off), because the HIP kernel is held to
 * bit-for-bit equality with this file on the same Philox stream.  Three
 * transcendental-free rewrites of reference expressions are used and are
 * mathematically equal to the originals:
 *   pow(1-cos,5)                     -> x*x, squared, times x       (cpu/material.h:93)
 *   sign(sin(10x)sin(10y)sin(10z))   -> parity of floor(10x/pi)+... (cpu/texture.hpp:37-43)
 *   sphere uv (acos/atan2)           -> not evaluated: no supported texture reads u,v
 *                                                                   (cpu/sphere.h:49-55)
 *
 * PINNING.  The reference ships no tests, golden images or fixtures (SURVEY.md
 * section 4), so this file is pinned against the reference ITSELF: oracle/_ref is
 * cmake-cpu-version's own main.cpp compiled here with rand() hooked to the same
 * Philox stream (oracle/ref_harness.cpp, oracle/Makefile).  tests/test_oracle_pin.py
 * compares per-sample radiance and images, and tests/golden/ holds vectors the
 * hooked reference produced (generator: tests/golden/make_golden.py).  The
 * CUDA-only features (rects, cylinders, emission, constant background) have no
 * runnable reference here: for those this file is "parity unpinned" and is
 * checked by analytic known-answer tests instead (tests/test_primitives.py).
 */
#include "rt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ RNG
 * Replaces cpu/rtweekend.h:17-29 random_double() (one global rand() stream) and
 * gpu/rtweekend.cuh:23-29 (curand XORWOW per pixel).  Philox4x32-10 with key = seed and
 * counter = (pixel id, sample index, 0, 0) seeds a xorshift128 stream per sample; a
 * uniform is the top 24 bits of a word.
 * Restated from the published algorithm (Salmon, Moraes, Dror, Shaw: "Parallel
 * random numbers: as easy as 1, 2, 3", SC'11); known-answer vectors from the
 * Random123 distribution are checked in tests/test_philox.py. */
static void philox_block(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                         uint32_t k1, uint32_t out[4]) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0, c1 = n1, c2 = n2, c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0, out[1] = c1, out[2] = c2, out[3] = c3;
}

void rto_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    philox_block(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], out);
}

/* Each (pixel, sample) owns a xorshift128 generator (Marsaglia, "Xorshift RNGs", JSS 2003,
 * xor128) whose 128-bit state is one Philox block keyed by the seed. */
typedef struct rng_t {
    uint32_t x, y, z, w;
    uint64_t draws;
} rng_t;

static void rng_init(rng_t *g, uint64_t seed, uint32_t pixel, uint32_t sample) {
    uint32_t b[4];
    philox_block(pixel, sample, 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), b);
    g->x = b[0], g->y = b[1], g->z = b[2], g->w = b[3];
    if ((g->x | g->y | g->z | g->w) == 0u) g->w = 0x9E3779B9u; /* all-zero is a fixed point */
    g->draws = 0;
}

static uint32_t rng_word(rng_t *g) {
    uint32_t t = g->x ^ (g->x << 11);
    g->x = g->y, g->y = g->z, g->z = g->w;
    g->w = g->w ^ (g->w >> 19) ^ (t ^ (t >> 8));
    return g->w;
}

/* random_double(), cpu/rtweekend.h:17-25: uniform in [0,1) */
static float random_float(rng_t *g) {
    g->draws++;
    return (float)(rng_word(g) >> 8) * (1.0f / 16777216.0f);
}

/* random_double(min,max), cpu/rtweekend.h:27-29, for (-1,1): -1 + 2*xi (exact) */
static float random_pm1(rng_t *g) { return -1.0f + 2.0f * random_float(g); }

void rto_sample_stream(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t *out, int n) {
    rng_t g;
    rng_init(&g, seed, pixel, sample);
    for (int i = 0; i < n; ++i) out[i] = rng_word(&g);
}

/* ------------------------------------------------------------------ vec3
 * cpu/vec3.h:9-118 (class vec3 and free operators), fp32 */
typedef struct vec3 {
    float x, y, z;
} vec3;

static inline vec3 v3(float x, float y, float z) {
    vec3 r = {x, y, z};
    return r;
}
static inline vec3 vadd(vec3 a, vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline vec3 vsub(vec3 a, vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline vec3 vmul(vec3 a, vec3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline vec3 vscale(float t, vec3 a) { return v3(t * a.x, t * a.y, t * a.z); }
static inline vec3 vneg(vec3 a) { return v3(-a.x, -a.y, -a.z); }
/* a + t*b, one fma per component */
static inline vec3 vfma(float t, vec3 b, vec3 a) {
    return v3(fmaf(t, b.x, a.x), fmaf(t, b.y, a.y), fmaf(t, b.z, a.z));
}
/* dot(), cpu/vec3.h:106-109: x*x' + y*y' + z*z' as fma(x,x', fma(y,y', z*z')) */
static inline float dot(vec3 a, vec3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
static inline float length_squared(vec3 a) { return dot(a, a); }
/* unit_vector(v) = v / v.length() = (1/len) * v, cpu/vec3.h:102-104,118 */
static inline vec3 unit_vector(vec3 a) {
    float inv = 1.0f / sqrtf(length_squared(a));
    return vscale(inv, a);
}
/* vec3::near_zero, cpu/vec3.h:63-66 */
static inline int near_zero(vec3 a) {
    const float eps = 1e-8f;
    return fabsf(a.x) < eps && fabsf(a.y) < eps && fabsf(a.z) < eps;
}

/* random_in_unit_sphere(), cpu/vec3.h:121-129: rejection from (-1,1)^3, draws x,y,z */
static vec3 random_in_unit_sphere(rng_t *g) {
    for (;;) {
        float x = random_pm1(g);
        float y = random_pm1(g);
        float z = random_pm1(g);
        vec3 p = v3(x, y, z);
        if (length_squared(p) >= 1.0f) continue;
        return p;
    }
}
/* random_unit_vector(), cpu/vec3.h:131-134 */
static vec3 random_unit_vector(rng_t *g) { return unit_vector(random_in_unit_sphere(g)); }

/* reflect(), cpu/vec3.h:144-147: v - 2*dot(v,n)*n */
static inline vec3 reflect(vec3 v, vec3 n) {
    float k = 2.0f * dot(v, n);
    return vfma(-k, n, v);
}
/* refract(), cpu/vec3.h:149-155 */
static inline vec3 refract(vec3 uv, vec3 n, float etai_over_etat) {
    float cos_theta = fminf(-dot(uv, n), 1.0f);
    vec3 perp = vscale(etai_over_etat, vfma(cos_theta, n, uv));
    float k = -sqrtf(fabsf(1.0f - length_squared(perp)));
    return vfma(k, n, perp);
}

/* ------------------------------------------------------------------ ray, hit_record
 * cpu/ray.h:5-19, cpu/hittable.h:8-21.  `a`/`inv_a` cache direction.length_squared()
 * and its reciprocal, which every sphere::hit recomputes (cpu/sphere.h:17). */
typedef struct ray_t {
    vec3 orig, dir;
    float a, inv_a;
} ray_t;

static inline ray_t make_ray(vec3 o, vec3 d) {
    ray_t r;
    r.orig = o;
    r.dir = d;
    r.a = length_squared(d);
    r.inv_a = 1.0f / r.a;
    return r;
}
static inline vec3 ray_at(const ray_t *r, float t) { return vfma(t, r->dir, r->orig); }

typedef struct hit_record {
    vec3 p, normal;
    int material;
    float t;
    int front_face;
    /* rec.u, rec.v (gpu/hittable.cuh:12-13) are evaluated where a texture reads them (hit_uv below): only image
     * textures do, and get_sphere_uv's acos / atan2 per candidate hit would dominate sphere::hit.  What they are
     * computed from: */
    const struct rto_prim *prim;
} hit_record;

/* hit_record::set_face_normal, cpu/hittable.h:17-20 */
static inline void set_face_normal(hit_record *rec, const ray_t *r, vec3 outward) {
    rec->front_face = dot(r->dir, outward) < 0.0f;
    rec->normal = rec->front_face ? outward : vneg(outward);
}

/* ------------------------------------------------------------------ primitives */

/* sphere::hit, cpu/sphere.h:14-42 (= gpu/object.cuh:47-75).
 * root = (-hb -/+ sqrtd) / a is evaluated as a multiply by inv_a. */
static int sphere_hit(const rto_prim *sp, const ray_t *r, float t_min, float t_max,
                      hit_record *rec) {
    vec3 center = v3(sp->f[0], sp->f[1], sp->f[2]);
    float radius = sp->f[3];
    vec3 oc = vsub(r->orig, center);
    float hb = dot(oc, r->dir);
    float c = fmaf(oc.x, oc.x, fmaf(oc.y, oc.y, fmaf(oc.z, oc.z, -(radius * radius))));
    float disc = fmaf(hb, hb, -(r->a * c));
    if (disc < 0.0f) return 0;
    float sqrtd = sqrtf(disc);
    float root = (-hb - sqrtd) * r->inv_a;
    if (root < t_min || t_max < root) {
        root = (-hb + sqrtd) * r->inv_a;
        if (root < t_min || t_max < root) return 0;
    }
    rec->t = root;
    rec->p = ray_at(r, root);
    vec3 normal = vscale(1.0f / radius, vsub(rec->p, center)); /* (p - c) / r */
    set_face_normal(rec, r, normal);
    rec->material = sp->material;
    rec->prim = sp;
    return 1;
}

/* xy_rect/xz_rect/yz_rect::hit, gpu/object.cuh:105-122, 142-159, 175-192.
 * axis k: the rect lies in the plane coordinate[kaxis] = k; (a,b) are the other two. */
static inline float comp(vec3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }

static int rect_hit(const rto_prim *rc, const ray_t *r, float t_min, float t_max,
                    hit_record *rec) {
    int ka, aa, ba; /* plane axis, first and second in-plane axes */
    if (rc->type == 1) ka = 2, aa = 0, ba = 1;      /* xy_rect: z = k */
    else if (rc->type == 2) ka = 1, aa = 0, ba = 2; /* xz_rect: y = k */
    else ka = 0, aa = 1, ba = 2;                    /* yz_rect: x = k */
    float a0 = rc->f[0], a1 = rc->f[1], b0 = rc->f[2], b1 = rc->f[3], k = rc->f[4];
    float t = (k - comp(r->orig, ka)) / comp(r->dir, ka);
    if (t < t_min || t > t_max) return 0;
    float a = fmaf(t, comp(r->dir, aa), comp(r->orig, aa));
    float b = fmaf(t, comp(r->dir, ba), comp(r->orig, ba));
    if (a < a0 || a > a1 || b < b0 || b > b1) return 0;
    rec->t = t;
    vec3 outward = v3(ka == 0 ? 1.0f : 0.0f, ka == 1 ? 1.0f : 0.0f, ka == 2 ? 1.0f : 0.0f);
    /* dot(dir, unit axis) == dir[axis] exactly */
    rec->front_face = comp(r->dir, ka) < 0.0f;
    rec->normal = rec->front_face ? outward : vneg(outward);
    rec->material = rc->material;
    rec->p = ray_at(r, t);
    rec->prim = rc;
    return 1;
}

/* transform::apply_point / apply_vec / apply_normal, gpu/vec3.cuh:350-381, on the
 * 3x4 affine rows (w is always 1 for rotate/translate compositions) */
static inline vec3 xf_point(const float *m, vec3 p) {
    return v3(fmaf(m[0], p.x, fmaf(m[1], p.y, fmaf(m[2], p.z, m[3]))),
              fmaf(m[4], p.x, fmaf(m[5], p.y, fmaf(m[6], p.z, m[7]))),
              fmaf(m[8], p.x, fmaf(m[9], p.y, fmaf(m[10], p.z, m[11]))));
}
static inline vec3 xf_vec(const float *m, vec3 d) {
    return v3(fmaf(m[0], d.x, fmaf(m[1], d.y, m[2] * d.z)),
              fmaf(m[4], d.x, fmaf(m[5], d.y, m[6] * d.z)),
              fmaf(m[8], d.x, fmaf(m[9], d.y, m[10] * d.z)));
}

/* cylinder::hit + quadratic(), gpu/object.cuh:199-214, 233-290: open finite tube
 * about the object-space z axis */
static int cylinder_hit(const rto_prim *cy, const ray_t *r, float t_min, float t_max,
                        hit_record *rec) {
    float radius = cy->f[0], zmin = cy->f[1], zmax = cy->f[2];
    vec3 oo = xf_point(cy->m_inv, r->orig);
    vec3 od = xf_vec(cy->m_inv, r->dir);
    float a = fmaf(od.x, od.x, od.y * od.y);
    float b = 2.0f * fmaf(od.x, oo.x, od.y * oo.y);
    float c = fmaf(oo.x, oo.x, fmaf(oo.y, oo.y, -(radius * radius)));
    /* quadratic(): delta = b*b - 4*a*c */
    float delta = fmaf(b, b, -((4.0f * a) * c));
    if (delta < 0.0f) return 0;
    float sq = sqrtf(delta);
    float t0 = (-0.5f * (b - sq)) / a;
    float t1 = (-0.5f * (b + sq)) / a;
    if (t0 > t1) {
        float tmp = t0;
        t0 = t1;
        t1 = tmp;
    }
    if (t0 > t_max || t1 < t_min) return 0;
    float t = t0;
    if (t0 < t_min) {
        t = t1;
        if (t > t_max) return 0;
    }
    vec3 op = vfma(t, od, oo);
    if (op.z < zmin || op.z > zmax) {
        if (t == t1) return 0;
        t = t1;
        if (t > t_max || t < t_min) return 0;
        op = vfma(t, od, oo);
        if (op.z < zmin || op.z > zmax) return 0;
    }
    /* vec3(x,y,0).normalize(): componentwise division by the length */
    float len = sqrtf(fmaf(op.x, op.x, op.y * op.y));
    float nx = op.x / len, ny = op.y / len;
    rec->p = xf_point(cy->m, op);
    /* apply_normal: transpose of m_inv times (nx, ny, 0) */
    const float *mi = cy->m_inv;
    vec3 wn = v3(fmaf(mi[0], nx, mi[4] * ny), fmaf(mi[1], nx, mi[5] * ny),
                 fmaf(mi[2], nx, mi[6] * ny));
    set_face_normal(rec, r, wn);
    rec->material = cy->material;
    rec->t = t;
    rec->prim = cy;
    return 1;
}

/* ------------------------------------------------------------------ triangle
 * hit_triangle, taichi-version/hittable.py:38-71.  cross(a, b) with one fused multiply-add per component. */
static inline vec3 cross(vec3 a, vec3 b) {
    return v3(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}

/* the plane point r_interact and the ray parameter (hittable.py:44-52, 61) */
static int triangle_plane(const rto_prim *tr, const ray_t *r, vec3 *ri, float *root) {
    vec3 v1 = v3(tr->m[0], tr->m[1], tr->m[2]);
    vec3 n = v3(tr->m[9], tr->m[10], tr->m[11]);
    vec3 oc = vsub(r->orig, v1);
    float ocn = dot(oc, n);
    if (ocn < 0.0f) n = vneg(n), ocn = -ocn; /* n = -n: oc.dot(n) changes sign exactly */
    float a = sqrtf(r->a);                   /* ray_direction.norm() */
    float theta = dot(r->dir, n) / a;
    if (!(theta < 0.0f)) return 0;
    *ri = v3(r->orig.x - ((r->dir.x / a) * ocn) / theta, r->orig.y - ((r->dir.y / a) * ocn) / theta,
             r->orig.z - ((r->dir.z / a) * ocn) / theta);
    *root = ((-ocn) / theta) / a;
    return 1;
}

static int triangle_hit(const rto_prim *tr, const ray_t *r, float t_min, float t_max, hit_record *rec) {
    vec3 v1 = v3(tr->m[0], tr->m[1], tr->m[2]), v2 = v3(tr->m[3], tr->m[4], tr->m[5]), v3_ = v3(tr->m[6], tr->m[7], tr->m[8]);
    vec3 ri;
    float root;
    if (!triangle_plane(tr, r, &ri, &root)) return 0;
    if (root < t_min || root > t_max) return 0;
    vec3 e21 = vsub(v2, v1), e31 = vsub(v3_, v1), e32 = vsub(v3_, v2), e12 = vneg(e21);
    vec3 a1 = vsub(ri, v1), a2 = vsub(ri, v2);
    float n1 = dot(cross(a1, e21), cross(e31, e21));
    float n2 = dot(cross(a2, e12), cross(e32, e12));
    float n3 = dot(cross(a1, e31), cross(e21, e31));
    float n4 = dot(cross(a2, e32), cross(e12, e32));
    if (!(n1 > 0.0f && n2 > 0.0f && n3 > 0.0f && n4 > 0.0f)) return 0;
    rec->t = root;
    rec->p = ray_at(r, root);
    set_face_normal(rec, r, v3(tr->m[9], tr->m[10], tr->m[11])); /* hittable.py:254-259 */
    rec->material = tr->material;
    rec->prim = tr;
    return 1;
}

/* ------------------------------------------------------------------ texture coordinates
 * atan2 / acos with a fixed fp32 operation sequence (the HIP kernel's csrc/rt_trig.h, operation for operation):
 * Cephes atanf's reduction and polynomial. */
static float atan_unit(float a) {
    float y0 = 0.0f, t = a;
    if (a > 0.4142135679721832275390625f) {
        y0 = 0.785398185253143310546875f;
        t = (a - 1.0f) / (a + 1.0f);
    }
    float z = t * t;
    float p = fmaf(8.05374449538e-2f, z, -1.38776856032e-1f);
    p = fmaf(p, z, 1.99777106478e-1f);
    p = fmaf(p, z, -3.33329491539e-1f);
    return y0 + fmaf(p * z, t, t);
}
float rto_atan2f(float y, float x) {
    float ax = fabsf(x), ay = fabsf(y);
    float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    float r = 0.0f;
    if (mx > 0.0f) r = atan_unit(mn / mx);
    if (ay > ax) r = 1.57079637050628662109375f - r;
    if (x < 0.0f) r = 3.1415927410125732421875f - r;
    return y < 0.0f ? -r : r;
}
float rto_acosf(float c) {
    float s = sqrtf(fmaxf((1.0f - c) * (1.0f + c), 0.0f));
    return rto_atan2f(s, c);
}

/* rec.u, rec.v of the accepted hit: sphere gpu/object.cuh:87-93 (get_sphere_uv of the OUTWARD normal), rects
 * :113-114 / 150-151 / 183-184, cylinder :283-288 (object space), triangle taichi hittable.py:54-58, 233 */
static void hit_uv(const hit_record *rec, const ray_t *r, float *u, float *v) {
    const rto_prim *p = rec->prim;
    const float pi = 3.1415927410125732421875f;
    switch (p->type) {
    case 0: {
        vec3 on = rec->front_face ? rec->normal : vneg(rec->normal);
        float theta = rto_acosf(-on.y);
        float phi = rto_atan2f(-on.z, on.x) + pi;
        *u = phi / 6.283185482025146484375f;
        *v = theta / pi;
        break;
    }
    case 1:
    case 2:
    case 3: {
        float pa = p->type == 3 ? rec->p.y : rec->p.x, pb = p->type == 1 ? rec->p.y : rec->p.z;
        *u = (pa - p->f[0]) / (p->f[1] - p->f[0]);
        *v = (pb - p->f[2]) / (p->f[3] - p->f[2]);
        break;
    }
    case 4: {
        vec3 oo = xf_point(p->m_inv, r->orig), od = xf_vec(p->m_inv, r->dir);
        vec3 op = vfma(rec->t, od, oo);
        float phi = rto_atan2f(op.y, op.x) + 6.283185482025146484375f;
        *u = phi / 12.56637096405029296875f;
        *v = (op.z - p->f[1]) / (p->f[2] - p->f[1]);
        break;
    }
    default: {
        vec3 v1 = v3(p->m[0], p->m[1], p->m[2]), v2 = v3(p->m[3], p->m[4], p->m[5]), v3_ = v3(p->m[6], p->m[7], p->m[8]);
        vec3 ri;
        float root;
        triangle_plane(p, r, &ri, &root);
        vec3 a1 = vsub(ri, v1), a2 = vsub(ri, v2), a3 = vsub(ri, v3_);
        float w1 = sqrtf(length_squared(cross(a1, a2))) / sqrtf(length_squared(cross(vsub(v3_, v1), vsub(v3_, v2))));
        float w2 = sqrtf(length_squared(cross(a1, a3))) / sqrtf(length_squared(cross(vsub(v2, v1), vsub(v2, v3_))));
        float w3 = sqrtf(length_squared(cross(a3, a2))) / sqrtf(length_squared(cross(vsub(v1, v3_), vsub(v1, v2))));
        const float *t = p->m_inv; /* u1, u2, u3 */
        *u = fmaf(t[4], w3, fmaf(t[2], w2, t[0] * w1));
        *v = fmaf(t[5], w3, fmaf(t[3], w2, t[1] * w1));
        break;
    }
    }
}

/* hittable_list::hit, cpu/hittable_list.h:23-37 (= gpu/object.cuh:23-37) */
static int world_hit(const rto_scene *s, const ray_t *r, float t_min, float t_max,
                     hit_record *rec, rto_counts *cnt) {
    hit_record temp_rec;
    int hit_anything = 0;
    float closest_so_far = t_max;
    for (int i = 0; i < s->num_prims; ++i) {
        const rto_prim *p = &s->prims[i];
        int h;
        switch (p->type) {
        case 0: h = sphere_hit(p, r, t_min, closest_so_far, &temp_rec); break;
        case 1:
        case 2:
        case 3: h = rect_hit(p, r, t_min, closest_so_far, &temp_rec); break;
        case 4: h = cylinder_hit(p, r, t_min, closest_so_far, &temp_rec); break;
        default: h = triangle_hit(p, r, t_min, closest_so_far, &temp_rec); break;
        }
        if (h) {
            hit_anything = 1;
            closest_so_far = temp_rec.t;
            *rec = temp_rec;
        }
    }
    if (cnt) {
        cnt->queries++;
        cnt->prim_tests += (uint64_t)s->num_prims;
    }
    return hit_anything;
}

/* ------------------------------------------------------------------ textures
 * solid_color::value cpu/texture.hpp:11-26; checker_texture::value :28-49.
 * sin(t) < 0  <=>  floor(t/pi) odd (t != 0), so the sign of the triple product is
 * the parity of the three floors; a zero factor makes the product 0 -> "even". */
static vec3 texture_value(const rto_scene *s, int tex, const hit_record *rec, const ray_t *r) {
    const rto_texture *t = &s->texs[tex];
    const vec3 p = rec->p;
    vec3 c0 = v3(t->c0[0], t->c0[1], t->c0[2]);
    if (t->type == 0) return c0;
    if (t->type == 2) { /* taichi-version/material.py:137-144: texel[int(frac(u) rows)][int(frac(v) cols)] / 255 */
        const rto_image *im = &s->images[(int)t->c0[0]];
        float u, v;
        hit_uv(rec, r, &u, &v);
        int x = (int)((u - floorf(u)) * (float)im->rows), y = (int)((v - floorf(v)) * (float)im->cols);
        if (x > im->rows - 1) x = im->rows - 1;
        if (y > im->cols - 1) y = im->cols - 1;
        const uint8_t *px = im->rgb + ((size_t)x * im->cols + y) * 3;
        return v3((float)px[0] / 255.0f, (float)px[1] / 255.0f, (float)px[2] / 255.0f);
    }
    const float inv_pi = 0.318309886183790671538f;
    float tx = 10.0f * p.x, ty = 10.0f * p.y, tz = 10.0f * p.z;
    int kx = (int)floorf(tx * inv_pi), ky = (int)floorf(ty * inv_pi), kz = (int)floorf(tz * inv_pi);
    int zero = (tx == 0.0f) || (ty == 0.0f) || (tz == 0.0f);
    int odd = !zero && (((kx + ky + kz) & 1) != 0);
    return odd ? v3(t->c1[0], t->c1[1], t->c1[2]) : c0;
}

/* ------------------------------------------------------------------ materials */

/* lambertian::scatter, cpu/material.h:25-35 */
static int lambertian_scatter(const rto_scene *s, const rto_material *m, const ray_t *r_in, const hit_record *rec,
                              vec3 *attenuation, ray_t *scattered, rng_t *g) {
    vec3 dir = vadd(rec->normal, random_unit_vector(g));
    if (near_zero(dir)) dir = rec->normal;
    *scattered = make_ray(rec->p, dir);
    *attenuation = texture_value(s, m->texture, rec, r_in);
    return 1;
}

/* metal::scatter, cpu/material.h:47-53 (draws the fuzz sample even when fuzz == 0) */
static int metal_scatter(const rto_material *m, const ray_t *r_in, const hit_record *rec,
                         vec3 *attenuation, ray_t *scattered, rng_t *g) {
    vec3 reflected = reflect(unit_vector(r_in->dir), rec->normal);
    vec3 fz = random_in_unit_sphere(g);
    vec3 dir = vfma(m->fuzz, fz, reflected);
    *scattered = make_ray(rec->p, dir);
    *attenuation = v3(m->albedo[0], m->albedo[1], m->albedo[2]);
    return dot(dir, rec->normal) > 0.0f;
}

/* dielectric::scatter + reflectance, cpu/material.h:66-95.  The uniform is drawn
 * only when refraction is possible (|| short-circuit, :77). */
static int dielectric_scatter(const rto_material *m, const ray_t *r_in, const hit_record *rec,
                              vec3 *attenuation, ray_t *scattered, rng_t *g) {
    *attenuation = v3(1.0f, 1.0f, 1.0f);
    float ratio = rec->front_face ? (1.0f / m->ir) : m->ir;
    vec3 ud = unit_vector(r_in->dir);
    float cos_theta = fminf(-dot(ud, rec->normal), 1.0f);
    float sin_theta = sqrtf(fmaf(-cos_theta, cos_theta, 1.0f));
    int cannot_refract = ratio * sin_theta > 1.0f;
    int do_reflect = cannot_refract;
    if (!cannot_refract) {
        float r0 = (1.0f - ratio) / (1.0f + ratio);
        r0 = r0 * r0;
        float x = 1.0f - cos_theta;
        float x2 = x * x;
        float x5 = (x2 * x2) * x;
        float refl = fmaf(1.0f - r0, x5, r0);
        do_reflect = refl > random_float(g);
    }
    vec3 dir = do_reflect ? reflect(ud, rec->normal) : refract(ud, rec->normal, ratio);
    *scattered = make_ray(rec->p, dir);
    return 1;
}

/* ------------------------------------------------------------------ camera
 * camera::camera cpu/camera.h:9-31, in fp64 then rounded once */
void rto_derive_camera(const rto_camera_params *p, rto_camera *out) {
    const double pi = acos(-1.0);
    double theta = p->vfov * pi / 180.0;
    double h = tan(theta / 2);
    double vh = 2.0 * h, vw = p->aspect * vh;
    double w[3], u[3], v[3], len;
    for (int i = 0; i < 3; ++i) w[i] = p->lookfrom[i] - p->lookat[i];
    len = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    for (int i = 0; i < 3; ++i) w[i] *= 1.0 / len;
    u[0] = p->vup[1] * w[2] - p->vup[2] * w[1];
    u[1] = p->vup[2] * w[0] - p->vup[0] * w[2];
    u[2] = p->vup[0] * w[1] - p->vup[1] * w[0];
    len = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    for (int i = 0; i < 3; ++i) u[i] *= 1.0 / len;
    v[0] = w[1] * u[2] - w[2] * u[1];
    v[1] = w[2] * u[0] - w[0] * u[2];
    v[2] = w[0] * u[1] - w[1] * u[0];
    for (int i = 0; i < 3; ++i) {
        double hor = p->focus_dist * vw * u[i];
        double ver = p->focus_dist * vh * v[i];
        out->origin[i] = (float)p->lookfrom[i];
        out->horizontal[i] = (float)hor;
        out->vertical[i] = (float)ver;
        out->lower_left[i] = (float)(p->lookfrom[i] - hor / 2 - ver / 2 - p->focus_dist * w[i]);
        out->u[i] = (float)u[i];
        out->v[i] = (float)v[i];
        out->w[i] = (float)w[i];
    }
    out->lens_radius = (float)(p->aperture / 2);
}

/* camera::get_ray, cpu/camera.h:32-39 with random_in_unit_disk cpu/vec3.h:157-165.
 * The disk sample is drawn whenever blur is enabled, also for lens_radius == 0
 * (as the reference does); with blur disabled (gpu/camera.cuh:33-34) no draw. */
static ray_t camera_get_ray(const rto_scene *sc, float s, float t, rng_t *g) {
    const rto_camera *c = &sc->cam;
    vec3 origin = v3(c->origin[0], c->origin[1], c->origin[2]);
    vec3 llc = v3(c->lower_left[0], c->lower_left[1], c->lower_left[2]);
    vec3 hor = v3(c->horizontal[0], c->horizontal[1], c->horizontal[2]);
    vec3 ver = v3(c->vertical[0], c->vertical[1], c->vertical[2]);
    vec3 offset = v3(0.0f, 0.0f, 0.0f);
    if (sc->flags & RTO_FLAG_DEFOCUS_BLUR) {
        float px, py;
        for (;;) {
            px = random_pm1(g);
            py = random_pm1(g);
            if (fmaf(px, px, py * py) >= 1.0f) continue;
            break;
        }
        float rdx = c->lens_radius * px, rdy = c->lens_radius * py;
        vec3 cu = v3(c->u[0], c->u[1], c->u[2]), cv = v3(c->v[0], c->v[1], c->v[2]);
        offset = v3(fmaf(cu.x, rdx, cv.x * rdy), fmaf(cu.y, rdx, cv.y * rdy),
                    fmaf(cu.z, rdx, cv.z * rdy));
    }
    /* lower_left + s*horizontal + t*vertical - origin - offset, left to right */
    vec3 d = vfma(s, hor, llc);
    d = vfma(t, ver, d);
    d = vsub(d, origin);
    d = vsub(d, offset);
    return make_ray(vadd(origin, offset), d);
}

/* ------------------------------------------------------------------ integrator
 * ray_color: cpu/main.cpp:13-43 generalised with gpu/main.cu:17-70's emission and
 * constant background.  With no emissive material and the sky gradient this IS the
 * CPU function: L stays 0 until the miss, where L = 0 + beta*sky = sky*ret; an
 * absorbed or depth-exhausted path returns 0. */
/* debugging aid for the tests: when set, ray_color records every query of the calling thread
 * as 8 floats (origin, direction, hit t or -1, hit flag) */
static __thread float *g_trace;
static __thread int g_trace_n, g_trace_max;

static vec3 ray_color(const rto_scene *s, ray_t now, int depth, rng_t *g, rto_counts *cnt,
                      int *queries) {
    vec3 beta = v3(1.0f, 1.0f, 1.0f); /* `ret` / accumulated_attenuation */
    vec3 L = v3(0.0f, 0.0f, 0.0f);    /* accumulated_color */
    while (depth > 0) {
        hit_record rec;
        memset(&rec, 0, sizeof rec);
        /* Russian roulette, 朴素光线追踪/4_0_path_tracing.py:45-46: `if ti.random() > p_RR: break` before
         * every query (the path returns what it has collected).  A survivor's throughput is divided
         * by p at once, which compensates everything gathered from this query on; the reference
         * divides after the scatter (:88), which leaves the segment that ends at a light or at the
         * background uncompensated -- its images are darker by the factor p. */
        if (s->rr_p > 0.0f) {
            if (random_float(g) > s->rr_p) return L;
            beta = v3(beta.x / s->rr_p, beta.y / s->rr_p, beta.z / s->rr_p);
        }
        ++*queries;
        const int did_hit = world_hit(s, &now, 0.001f, INFINITY, &rec, cnt);
        if (g_trace && g_trace_n < g_trace_max) {
            float *t = g_trace + 8 * g_trace_n++;
            t[0] = now.orig.x, t[1] = now.orig.y, t[2] = now.orig.z;
            t[3] = now.dir.x, t[4] = now.dir.y, t[5] = now.dir.z;
            t[6] = did_hit ? rec.t : -1.0f, t[7] = (float)did_hit;
        }
        if (did_hit) {
            const rto_material *m = &s->mats[rec.material];
            ray_t scattered;
            vec3 attenuation;
            int did_scatter;
            if (cnt) {
                cnt->hits++;
                cnt->scatter[m->type]++;
            }
            switch (m->type) {
            case 0: did_scatter = lambertian_scatter(s, m, &now, &rec, &attenuation, &scattered, g); break;
            case 1: did_scatter = metal_scatter(m, &now, &rec, &attenuation, &scattered, g); break;
            case 2: did_scatter = dielectric_scatter(m, &now, &rec, &attenuation, &scattered, g); break;
            default: { /* diffuse_light, gpu/material.cuh:161-182: emits, never scatters */
                vec3 e = texture_value(s, m->texture, &rec, &now);
                L = v3(fmaf(e.x, beta.x, L.x), fmaf(e.y, beta.y, L.y), fmaf(e.z, beta.z, L.z));
                did_scatter = 0;
                break;
            }
            }
            if (did_scatter) {
                beta = vmul(beta, attenuation);
                depth--;
                now = scattered;
                continue;
            }
            return L; /* absorbed: cpu/main.cpp:32 returns 0 = L when nothing emits */
        }
        if (cnt) cnt->misses++;
        vec3 bg;
        if (s->flags & RTO_FLAG_SKY_GRADIENT) { /* cpu/main.cpp:36-38 */
            vec3 ud = unit_vector(now.dir);
            float t = 0.5f * (ud.y + 1.0f);
            float omt = 1.0f - t;
            bg = v3(fmaf(t, 0.5f, omt), fmaf(t, 0.7f, omt), fmaf(t, 1.0f, omt));
        } else { /* gpu/main.cu:63 */
            bg = v3(s->background[0], s->background[1], s->background[2]);
        }
        return v3(fmaf(beta.x, bg.x, L.x), fmaf(beta.y, bg.y, L.y), fmaf(beta.z, bg.z, L.z));
    }
    return L; /* depth exhausted: cpu/main.cpp:42 (0), gpu/main.cu:69 (emission so far) */
}

/* one sample of render(), cpu/main.cpp:48-53 */
int rto_sample(const rto_scene *s, uint64_t seed, int x, int y, int sample, float rgb[3],
               rto_counts *counts) {
    rng_t g;
    rng_init(&g, seed, (uint32_t)(y * s->width + x), (uint32_t)sample);
    /* u = (i + xi) / (W - 1), cpu/main.cpp:49-50: a multiply by the fp32 reciprocal (equal in real arithmetic; the HIP
     * kernel saves two IEEE divisions per sample with it) */
    float u = ((float)x + random_float(&g)) * (1.0f / (float)(s->width - 1));
    float v = ((float)y + random_float(&g)) * (1.0f / (float)(s->height - 1));
    ray_t r = camera_get_ray(s, u, v, &g);
    int queries = 0;
    vec3 c = ray_color(s, r, s->max_depth, &g, counts, &queries);
    rgb[0] = c.x, rgb[1] = c.y, rgb[2] = c.z;
    if (counts) {
        counts->samples++;
        counts->rng_draws += g.draws;
    }
    return queries;
}

/* rto_sample + the list of its closest-hit queries (8 floats each, see g_trace); returns the
 * number of queries recorded */
int rto_trace_sample(const rto_scene *s, uint64_t seed, int x, int y, int sample, float rgb[3],
                     float *queries8, int max_queries) {
    g_trace = queries8, g_trace_n = 0, g_trace_max = max_queries;
    rto_sample(s, seed, x, y, sample, rgb, 0);
    g_trace = 0;
    return g_trace_n;
}

int rto_hit_uv(const rto_scene *s, const float o[3], const float d[3], float uv[2], float *t, int *prim) {
    ray_t r = make_ray(v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2]));
    hit_record rec;
    memset(&rec, 0, sizeof rec);
    if (!world_hit(s, &r, 0.001f, INFINITY, &rec, 0)) return 0;
    hit_uv(&rec, &r, &uv[0], &uv[1]);
    if (t) *t = rec.t;
    if (prim) *prim = (int)(rec.prim - s->prims);
    return 1;
}

static void counts_add(rto_counts *a, const rto_counts *b) {
    a->samples += b->samples;
    a->queries += b->queries;
    a->prim_tests += b->prim_tests;
    a->hits += b->hits;
    a->misses += b->misses;
    for (int i = 0; i < 4; ++i) a->scatter[i] += b->scatter[i];
    a->rng_draws += b->rng_draws;
}

/* Pixel accumulation.  The reference adds sample radiances into a running sum
 * (res += ray_color(...), cpu/main.cpp:47,52: fp64; gpu/main.cu:100: fp32).  Here each
 * fp32 sample is converted to 64-bit fixed point with 24 fractional bits (round to nearest
 * even, |sample| clamped to 2^16, NaN -> 0) and summed in integers: exact, hence independent of
 * the order in which samples finish -- which lets the HIP kernel hand samples of a tile to
 * whichever lane is free.  (Resolution 2^-25 per sample: below the rounding of the one final
 * conversion to fp32 for every sum >= 1; 2^23 samples of the largest value cannot wrap the sum.) */
static uint64_t radiance_to_fixed(float v) {
    if (!(fabsf(v) <= 65536.0f)) v = (v != v) ? 0.0f : copysignf(65536.0f, v); /* NaN -> 0, clamp */
    return (uint64_t)llrint((double)v * 16777216.0);
}
static float fixed_to_sum(uint64_t t) { return (float)((double)(int64_t)t * (1.0 / 16777216.0)); }

/* render() per pixel, cpu/main.cpp:45-55, over the pixels [x0, x1) x [y0, y1) of the loop of
 * :99-106 (rgb_sum is the FULL image buffer; only the window is written).
 * rto_render below is the whole-rows form; its spp_chunk is accepted for interface symmetry with
 * rt_opts and has no effect on the result (the sum is exact). */
int rto_render_rect(const rto_scene *s, uint64_t seed, int x0, int x1, int y0, int y1, int sample_first,
                    int sample_count, float *rgb_sum, rto_counts *counts, int threads) {
    if (!s || !rgb_sum || y0 < 0 || y1 > s->height || y0 > y1 || x0 < 0 || x1 > s->width || x0 > x1 ||
        sample_count < 0)
        return 1;
    rto_counts total;
    memset(&total, 0, sizeof total);
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#else
    threads = 1;
#endif
#pragma omp parallel num_threads(threads)
    {
        rto_counts local;
        memset(&local, 0, sizeof local);
#pragma omp for schedule(dynamic, 1)
        for (int y = y0; y < y1; ++y) {
            for (int x = x0; x < x1; ++x) {
                uint64_t tot[3] = {0, 0, 0};
                for (int k = 0; k < sample_count; ++k) {
                    float rgb[3];
                    rto_sample(s, seed, x, y, sample_first + k, rgb, counts ? &local : 0);
                    tot[0] += radiance_to_fixed(rgb[0]);
                    tot[1] += radiance_to_fixed(rgb[1]);
                    tot[2] += radiance_to_fixed(rgb[2]);
                }
                float *o = rgb_sum + ((size_t)y * s->width + x) * 3;
                o[0] = fixed_to_sum(tot[0]), o[1] = fixed_to_sum(tot[1]), o[2] = fixed_to_sum(tot[2]);
            }
        }
#pragma omp critical
        counts_add(&total, &local);
    }
    if (counts) *counts = total;
    return 0;
}

/* whole rows [y0, y1): the pixel loop of cpu/main.cpp:99-106 restricted to a band */
int rto_render(const rto_scene *s, uint64_t seed, int y0, int y1, int sample_first,
               int sample_count, int spp_chunk, float *rgb_sum, rto_counts *counts, int threads) {
    (void)spp_chunk;
    if (!s) return 1;
    return rto_render_rect(s, seed, 0, s->width, y0, y1, sample_first, sample_count, rgb_sum, counts, threads);
}

/* Timing leg of bench.py's CPU baseline: this restatement on the rows y in [y0, y1) with
 * ((y - y0) % period) < band (every period-th band of `band` rows), one parallel loop over
 * (row, 64-pixel span).  Returns seconds; *checksum keeps the work alive. */
double rto_time_sample(const rto_scene *s, uint64_t seed, int y0, int y1, int period, int band, int spp,
                       int threads, double *checksum, long long *pixels) {
    if (!s || period <= 0 || band <= 0 || y0 < 0 || y1 > s->height) return -1.0;
    int nrows = 0;
    for (int y = y0; y < y1; ++y)
        if ((y - y0) % period < band) nrows++;
    const int spans = (s->width + 63) / 64;
    const long long jobs = (long long)nrows * spans;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
    double t0 = omp_get_wtime();
#else
    threads = 1;
    clock_t c0 = clock();
#endif
    double acc = 0.0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads) reduction(+ : acc)
    for (long long j = 0; j < jobs; ++j) {
        const int r = (int)(j / spans);                 /* r-th sampled row */
        const int y = y0 + (r / band) * period + r % band;
        const int xa = (int)(j % spans) * 64, xb = xa + 64 < s->width ? xa + 64 : s->width;
        for (int x = xa; x < xb; ++x) {
            uint64_t tot[3] = {0, 0, 0};
            for (int k = 0; k < spp; ++k) {
                float rgb[3];
                rto_sample(s, seed, x, y, k, rgb, 0);
                tot[0] += radiance_to_fixed(rgb[0]);
                tot[1] += radiance_to_fixed(rgb[1]);
                tot[2] += radiance_to_fixed(rgb[2]);
            }
            acc += (double)fixed_to_sum(tot[0]) + (double)fixed_to_sum(tot[1]) + (double)fixed_to_sum(tot[2]);
        }
    }
    if (checksum) *checksum = acc;
    if (pixels) *pixels = (long long)nrows * s->width;
#ifdef _OPENMP
    return omp_get_wtime() - t0;
#else
    return (double)(clock() - c0) / CLOCKS_PER_SEC;
#endif
}

/* aabb::hit, gpu/aabb.hpp:15-29 (slab test; t-range in double as the reference has it) */
int rto_aabb_hit(const float bmin[3], const float bmax[3], const float orig[3],
                 const float dir[3], float t_min_f, float t_max_f) {
    double t_min = t_min_f, t_max = t_max_f;
    for (int a = 0; a < 3; ++a) {
        float invD = 1.0f / dir[a];
        float t0 = (bmin[a] - orig[a]) * invD;
        float t1 = (bmax[a] - orig[a]) * invD;
        if (invD < 0.0f) {
            float tmp = t0;
            t0 = t1;
            t1 = tmp;
        }
        t_min = t0 > t_min ? t0 : t_min;
        t_max = t1 < t_max ? t1 : t_max;
        if (t_max <= t_min) return 0;
    }
    return 1;
}

/* write_color, gpu/color.cuh:70-95 (fp32 form of cpu/color.h:14-35);
 * gamma == 0: write_image's linear bytes, gpu/color.cuh:15-35 */
int rto_quantize(float sum, int spp, int gamma) {
    float v;
    if (gamma) {
        float scale = 1.0f / (float)spp;
        v = sqrtf(sum * scale);
    } else {
        v = sum / (float)spp;
    }
    if (v < 0.0f) v = 0.0f; /* clamp(), gpu/rtweekend.cuh:31-37 */
    if (v > 0.999f) v = 0.999f;
    return (int)(256.0f * v);
}

int rto_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
