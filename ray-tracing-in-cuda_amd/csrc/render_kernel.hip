// The hot path: one work-item owns one pixel (of one sample chunk) and runs the whole
// sample-and-bounce loop.  gfx950 (MI355X) only.
//
// Replaces   render<<<(W/8+1,H/8+1),(8,8)>>>   gpu-version/main.cu:72-105
//            ray_color                         gpu-version/main.cu:17-70  (semantics:
//                                              cmake-cpu-version/main.cpp:13-43)
//            hittable_list::hit + sphere/rect/cylinder::hit   gpu-version/object.cuh
//            material::scatter / emitted       gpu-version/material.cuh
//            camera::get_ray                   cmake-cpu-version/camera.h:32-39
//            curand XORWOW per-pixel state     -> stateless Philox4x32-10 (philox.h)
//
// Shape of the kernel
//   * 256-thread workgroup = 4 wave64; each wave owns an 8x8 pixel tile (coherent
//     primary rays), the workgroup a 32x8 strip of one row tile.
//   * the primitive tables the inner loop reads ("hot" part of the scene image,
//     device_scene.h) are copied into LDS once per workgroup; all lanes of a wave
//     read the same record each iteration, i.e. one broadcast ds_read_b128 per
//     sphere.  No virtual calls, no pointer chasing; cold data (1/r, material
//     records) stays in global memory and is read once per bounce.
//   * lanes stay converged across bounces: the loop body is one closest-hit query
//     for every live lane; a lane whose path ends (miss / absorbed / depth) adds the
//     radiance to its pixel sum and starts its NEXT sample in the same iteration, so
//     the wave-uniform primitive loop always runs with full occupancy; the loop
//     leaves on !__any(active).
//   * arithmetic: fp32, every fused multiply-add explicit (-ffp-contract=off), IEEE
//     sqrt and divide, so results are bit-identical to the scalar restatement the
//     tests check against.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "device_scene.h"
#include "philox.h"
#include "../../include/rtmi.h"

namespace rtmi {

static constexpr float kTMin = 0.001f;  // main.cu:45 / main.cpp:22

// ---------------------------------------------------------------- RNG
struct LaneRng {
    uint32_t pixel, sample, block;
    uint32_t b0, b1, b2, b3;
    int pos;
    uint32_t draws;
};

__device__ __forceinline__ void rng_start(LaneRng &g, uint32_t pixel, uint32_t sample) {
    g.pixel = pixel;
    g.sample = sample;
    g.block = 0;
    g.pos = 4;
}

template <bool COUNT>
__device__ __forceinline__ float rng_next(LaneRng &g, uint32_t k0, uint32_t k1) {
    if (g.pos == 4) {
        Philox4 p = philox4x32_10(g.pixel, g.sample, g.block, 0u, k0, k1);
        g.b0 = p.v[0], g.b1 = p.v[1], g.b2 = p.v[2], g.b3 = p.v[3];
        g.block++;
        g.pos = 0;
    }
    uint32_t w = g.pos == 0 ? g.b0 : (g.pos == 1 ? g.b1 : (g.pos == 2 ? g.b2 : g.b3));
    g.pos++;
    if (COUNT) g.draws++;
    return (float)(w >> 8) * (1.0f / 16777216.0f);
}

template <bool COUNT>
__device__ __forceinline__ float rng_pm1(LaneRng &g, uint32_t k0, uint32_t k1) {
    return -1.0f + 2.0f * rng_next<COUNT>(g, k0, k1);
}

__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) {
    return fmaf(ax, bx, fmaf(ay, by, az * bz));
}

// checker_texture::value, texture.cuh:44-52: sign of sin(10x)sin(10y)sin(10z) as the
// parity of floor(10x/pi) + floor(10y/pi) + floor(10z/pi); zero factor -> even
__device__ __forceinline__ bool checker_odd(float px, float py, float pz) {
    const float inv_pi = 0.318309886183790671538f;
    float tx = 10.0f * px, ty = 10.0f * py, tz = 10.0f * pz;
    int kx = (int)floorf(tx * inv_pi), ky = (int)floorf(ty * inv_pi), kz = (int)floorf(tz * inv_pi);
    bool zero = (tx == 0.0f) || (ty == 0.0f) || (tz == 0.0f);
    return !zero && (((kx + ky + kz) & 1) != 0);
}

// original list index of a grouped primitive id (cold tables), for the tie rule
__device__ __forceinline__ int list_index_of(const RenderParams &P, const float4 *__restrict__ image, int id) {
    if (id < P.ns) return __float_as_int(image[P.off_sph_cold + id].z);
    if (id < P.ns + P.nr) return __float_as_int(image[P.off_rect_cold + (id - P.ns)].y);
    return __float_as_int(image[P.off_cyl_cold + 4 * (id - P.ns - P.nr) + 3].y);
}

// ---------------------------------------------------------------- kernel
template <bool COUNT>
__global__ __launch_bounds__(256) void render_kernel(const RenderParams P, const float4 *__restrict__ image,
                                                     float *__restrict__ out, DevCounters *__restrict__ counters) {
    extern __shared__ float4 lds[];
    // stage the hot tables (hittable_list contents) into LDS
    for (int i = threadIdx.x; i < P.hot_vec4; i += 256) lds[i] = image[i];
    __syncthreads();
    float *stage = reinterpret_cast<float *>(lds + P.hot_vec4);  // 4 waves x 192 floats

    // workgroup -> (sample chunk, 8-row band of the shard, 32-pixel strip)
    int b = blockIdx.x;
    const int bx = b % P.blocks_x;
    b /= P.blocks_x;
    const int band = b % P.bands;
    const int chunk = b / P.bands;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int x = bx * 32 + wave * 8 + (lane & 7);
    const int lr = band * 8 + (lane >> 3);  // dense local row of this shard
    const int tl = lr / P.tile_rows;
    const int y = (P.tile_first + tl * P.tile_stride) * P.tile_rows + (lr - tl * P.tile_rows);
    const bool valid = x < P.width && lr < P.local_rows && y < P.height;

    const uint32_t k0 = P.seed_lo, k1 = P.seed_hi;
    const float4 *sph = lds;
    const float4 *rect = lds + P.off_rect_hot;
    const float4 *cyl = lds + P.off_cyl_hot;
    const int ns = P.ns, nr = P.nr, nc = P.nc;

    int s_next = P.sample_first + chunk * P.spp_chunk;
    int s_end = s_next + P.spp_chunk;
    if (s_end > P.sample_first + P.sample_count) s_end = P.sample_first + P.sample_count;
    if (!valid) s_end = s_next;

    const float fx = (float)x, fy = (float)y;
    const float wm1 = (float)(P.width - 1), hm1 = (float)(P.height - 1);
    const uint32_t pixel_id = (uint32_t)(y * P.width + x);

    LaneRng rng;
    rng.pixel = pixel_id, rng.sample = 0, rng.block = 0, rng.pos = 4, rng.draws = 0;
    rng.b0 = rng.b1 = rng.b2 = rng.b3 = 0;

    float sum_r = 0.0f, sum_g = 0.0f, sum_b = 0.0f;
    float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 1, ra = 1, rinv_a = 1;
    float beta_r = 1, beta_g = 1, beta_b = 1, L_r = 0, L_g = 0, L_b = 0;
    int depth = 0;
    bool active = false;

    uint32_t c_samples = 0, c_queries = 0, c_hits = 0, c_misses = 0;
    uint32_t c_scatter0 = 0, c_scatter1 = 0, c_scatter2 = 0, c_scatter3 = 0;

    for (;;) {
        // ---- refill: a lane without a live path starts its next sample
        // (render()'s sample loop, main.cu:95-101; camera::get_ray camera.h:32-39)
        if (!active && s_next < s_end) {
            rng_start(rng, pixel_id, (uint32_t)s_next);
            s_next++;
            float u = (fx + rng_next<COUNT>(rng, k0, k1)) / wm1;
            float v = (fy + rng_next<COUNT>(rng, k0, k1)) / hm1;
            float offx = 0.0f, offy = 0.0f, offz = 0.0f;
            if (P.flags & RT_FLAG_DEFOCUS_BLUR) {
                float px, py;
                do {  // random_in_unit_disk, vec3.h:157-165
                    px = rng_pm1<COUNT>(rng, k0, k1);
                    py = rng_pm1<COUNT>(rng, k0, k1);
                } while (fmaf(px, px, py * py) >= 1.0f);
                float rdx = P.cam.lens_radius * px, rdy = P.cam.lens_radius * py;
                offx = fmaf(P.cam.u[0], rdx, P.cam.v[0] * rdy);
                offy = fmaf(P.cam.u[1], rdx, P.cam.v[1] * rdy);
                offz = fmaf(P.cam.u[2], rdx, P.cam.v[2] * rdy);
            }
            dx = fmaf(v, P.cam.vertical[0], fmaf(u, P.cam.horizontal[0], P.cam.lower_left[0]));
            dy = fmaf(v, P.cam.vertical[1], fmaf(u, P.cam.horizontal[1], P.cam.lower_left[1]));
            dz = fmaf(v, P.cam.vertical[2], fmaf(u, P.cam.horizontal[2], P.cam.lower_left[2]));
            dx = (dx - P.cam.origin[0]) - offx;
            dy = (dy - P.cam.origin[1]) - offy;
            dz = (dz - P.cam.origin[2]) - offz;
            ox = P.cam.origin[0] + offx;
            oy = P.cam.origin[1] + offy;
            oz = P.cam.origin[2] + offz;
            ra = dot3(dx, dy, dz, dx, dy, dz);
            rinv_a = 1.0f / ra;
            beta_r = beta_g = beta_b = 1.0f;
            L_r = L_g = L_b = 0.0f;
            depth = P.max_depth;
            active = true;
            if (COUNT) c_samples++;
        }
        if (!__any(active)) break;  // every lane of the wave is out of samples

        if (active) {
            // ---- closest-hit query over the LDS-resident list (hittable_list::hit,
            // object.cuh:23-37).  Wave-uniform trip counts; `best_id` is the grouped id.
            float best_t = INFINITY;
            int best_id = -1;

            // spheres: sphere::hit, object.cuh:47-75.  Early-outs that need no sqrt:
            //   disc < 0                      -> no real root
            //   hb >= 0 and c >= 0            -> both roots <= 0 < t_min  (then
            //     sqrt(disc) <= sqrt(fl(hb*hb)) = hb, so (-hb + sqrtd) <= 0 exactly)
#pragma unroll 4
            for (int i = 0; i < ns; ++i) {
                const float4 s = sph[i];
                const float ocx = ox - s.x, ocy = oy - s.y, ocz = oz - s.z;
                const float hb = dot3(ocx, ocy, ocz, dx, dy, dz);
                const float cc = fmaf(ocx, ocx, fmaf(ocy, ocy, fmaf(ocz, ocz, -s.w)));
                const float disc = fmaf(hb, hb, -(ra * cc));
                if (!(disc < 0.0f) && !(hb >= 0.0f && cc >= 0.0f)) {
                    const float sq = sqrtf(disc);
                    float root = (-hb - sq) * rinv_a;
                    if (root < kTMin || best_t < root) root = (-hb + sq) * rinv_a;
                    if (!(root < kTMin || best_t < root)) {
                        best_t = root;
                        best_id = i;
                    }
                }
            }

            // axis-aligned rects: xy_rect/xz_rect/yz_rect::hit, object.cuh:105-192
            for (int j = 0; j < nr; ++j) {
                const float4 q0 = rect[2 * j], q1 = rect[2 * j + 1];
                const int axis = __float_as_int(q1.y);  // 0: z = k, 1: y = k, 2: x = k
                float ok, dk, oa, da, ob, db;
                if (axis == 0) ok = oz, dk = dz, oa = ox, da = dx, ob = oy, db = dy;
                else if (axis == 1) ok = oy, dk = dy, oa = ox, da = dx, ob = oz, db = dz;
                else ok = ox, dk = dx, oa = oy, da = dy, ob = oz, db = dz;
                const float t = (q1.x - ok) / dk;
                if (!(t < kTMin || t > best_t)) {
                    const float pa = fmaf(t, da, oa), pb = fmaf(t, db, ob);
                    if (!(pa < q0.x || pa > q0.y || pb < q0.z || pb > q0.w)) {
                        bool take = true;
                        if (t == best_t && best_id >= 0)  // tie: the later list entry wins
                            take = list_index_of(P, image, ns + j) > list_index_of(P, image, best_id);
                        if (take) {
                            best_t = t;
                            best_id = ns + j;
                        }
                    }
                }
            }

            // cylinders: cylinder::hit + quadratic, object.cuh:199-214, 233-290
            for (int k = 0; k < nc; ++k) {
                const float4 r0 = cyl[4 * k], r1 = cyl[4 * k + 1], r2 = cyl[4 * k + 2], pr = cyl[4 * k + 3];
                const float oox = fmaf(r0.x, ox, fmaf(r0.y, oy, fmaf(r0.z, oz, r0.w)));
                const float ooy = fmaf(r1.x, ox, fmaf(r1.y, oy, fmaf(r1.z, oz, r1.w)));
                const float ooz = fmaf(r2.x, ox, fmaf(r2.y, oy, fmaf(r2.z, oz, r2.w)));
                const float odx = fmaf(r0.x, dx, fmaf(r0.y, dy, r0.z * dz));
                const float ody = fmaf(r1.x, dx, fmaf(r1.y, dy, r1.z * dz));
                const float odz = fmaf(r2.x, dx, fmaf(r2.y, dy, r2.z * dz));
                const float qa = fmaf(odx, odx, ody * ody);
                const float qb = 2.0f * fmaf(odx, oox, ody * ooy);
                const float qc = fmaf(oox, oox, fmaf(ooy, ooy, -pr.x));
                const float delta = fmaf(qb, qb, -((4.0f * qa) * qc));
                if (!(delta < 0.0f)) {
                    const float sq = sqrtf(delta);
                    float t0 = (-0.5f * (qb - sq)) / qa;
                    float t1 = (-0.5f * (qb + sq)) / qa;
                    if (t0 > t1) {
                        const float tmp = t0;
                        t0 = t1;
                        t1 = tmp;
                    }
                    bool ok = !(t0 > best_t || t1 < kTMin);
                    float t = t0;
                    if (ok && t0 < kTMin) {
                        t = t1;
                        if (t > best_t) ok = false;
                    }
                    if (ok) {
                        float opz = fmaf(t, odz, ooz);
                        if (opz < pr.y || opz > pr.z) {
                            if (t == t1) ok = false;
                            else {
                                t = t1;
                                if (t > best_t || t < kTMin) ok = false;
                                else {
                                    opz = fmaf(t, odz, ooz);
                                    if (opz < pr.y || opz > pr.z) ok = false;
                                }
                            }
                        }
                    }
                    if (ok) {
                        bool take = true;
                        if (t == best_t && best_id >= 0)
                            take = list_index_of(P, image, ns + nr + k) > list_index_of(P, image, best_id);
                        if (take) {
                            best_t = t;
                            best_id = ns + nr + k;
                        }
                    }
                }
            }
            if (COUNT) c_queries++;

            // ---- shade the winner (ray_color body, main.cu:45-65 / main.cpp:22-38)
            bool path_done = false;
            if (best_id >= 0) {
                // hit record of the winner only (the reference fills one per candidate)
                float px, py, pz, nx, ny, nz;
                int mat;
                bool front;
                if (best_id < ns) {
                    const float4 s = sph[best_id];
                    const float4 cold = image[P.off_sph_cold + best_id];
                    px = fmaf(best_t, dx, ox), py = fmaf(best_t, dy, oy), pz = fmaf(best_t, dz, oz);
                    const float onx = cold.x * (px - s.x), ony = cold.x * (py - s.y), onz = cold.x * (pz - s.z);
                    front = dot3(dx, dy, dz, onx, ony, onz) < 0.0f;
                    nx = front ? onx : -onx, ny = front ? ony : -ony, nz = front ? onz : -onz;
                    mat = __float_as_int(cold.y);
                } else if (best_id < ns + nr) {
                    const int j = best_id - ns;
                    const int axis = __float_as_int(rect[2 * j + 1].y);
                    px = fmaf(best_t, dx, ox), py = fmaf(best_t, dy, oy), pz = fmaf(best_t, dz, oz);
                    const float dk = axis == 0 ? dz : (axis == 1 ? dy : dx);
                    front = dk < 0.0f;
                    // front ? (0,0,1) : -(0,0,1), zeros keep their sign as in the reference
                    const float sgn = front ? 1.0f : -1.0f, zer = front ? 0.0f : -0.0f;
                    nx = axis == 2 ? sgn : zer, ny = axis == 1 ? sgn : zer, nz = axis == 0 ? sgn : zer;
                    mat = __float_as_int(image[P.off_rect_cold + j].x);
                } else {
                    const int k = best_id - ns - nr;
                    const float4 r0 = cyl[4 * k], r1 = cyl[4 * k + 1], r2 = cyl[4 * k + 2];
                    const float4 *cc4 = image + P.off_cyl_cold + 4 * k;
                    const float4 m0 = cc4[0], m1 = cc4[1], m2 = cc4[2];
                    const float oox = fmaf(r0.x, ox, fmaf(r0.y, oy, fmaf(r0.z, oz, r0.w)));
                    const float ooy = fmaf(r1.x, ox, fmaf(r1.y, oy, fmaf(r1.z, oz, r1.w)));
                    const float ooz = fmaf(r2.x, ox, fmaf(r2.y, oy, fmaf(r2.z, oz, r2.w)));
                    const float odx = fmaf(r0.x, dx, fmaf(r0.y, dy, r0.z * dz));
                    const float ody = fmaf(r1.x, dx, fmaf(r1.y, dy, r1.z * dz));
                    const float odz = fmaf(r2.x, dx, fmaf(r2.y, dy, r2.z * dz));
                    const float opx = fmaf(best_t, odx, oox), opy = fmaf(best_t, ody, ooy), opz = fmaf(best_t, odz, ooz);
                    const float len = sqrtf(fmaf(opx, opx, opy * opy));
                    const float onx = opx / len, ony = opy / len;
                    px = fmaf(m0.x, opx, fmaf(m0.y, opy, fmaf(m0.z, opz, m0.w)));
                    py = fmaf(m1.x, opx, fmaf(m1.y, opy, fmaf(m1.z, opz, m1.w)));
                    pz = fmaf(m2.x, opx, fmaf(m2.y, opy, fmaf(m2.z, opz, m2.w)));
                    const float wnx = fmaf(r0.x, onx, r1.x * ony);
                    const float wny = fmaf(r0.y, onx, r1.y * ony);
                    const float wnz = fmaf(r0.z, onx, r1.z * ony);
                    front = dot3(dx, dy, dz, wnx, wny, wnz) < 0.0f;
                    nx = front ? wnx : -wnx, ny = front ? wny : -wny, nz = front ? wnz : -wnz;
                    mat = __float_as_int(cc4[3].x);
                }

                const float4 *M = image + P.off_mat + 3 * mat;
                const float4 q0 = M[0], q1 = M[1], q2 = M[2];
                const int kind = __float_as_int(q0.x);
                if (COUNT) {
                    c_hits++;
                    if (kind <= MK_LAMBERT_CHECKER) c_scatter0++;
                    else if (kind == MK_METAL) c_scatter1++;
                    else if (kind == MK_DIELECTRIC) c_scatter2++;
                    else c_scatter3++;
                }

                // random_in_unit_sphere (vec3.h:121-129) for every lane whose material
                // needs one (lambertian, metal): one converged rejection loop
                float sx = 0, sy = 0, sz = 0, sl2 = 1;
                if (kind <= MK_METAL) {
                    do {
                        sx = rng_pm1<COUNT>(rng, k0, k1);
                        sy = rng_pm1<COUNT>(rng, k0, k1);
                        sz = rng_pm1<COUNT>(rng, k0, k1);
                        sl2 = dot3(sx, sy, sz, sx, sy, sz);
                    } while (sl2 >= 1.0f);
                }

                float ndx, ndy, ndz;           // scattered direction
                float at_r, at_g, at_b;        // attenuation
                bool scattered = true;
                if (kind <= MK_LAMBERT_CHECKER) {  // lambertian::scatter, material.h:25-35
                    const float inv = 1.0f / sqrtf(sl2);
                    ndx = nx + inv * sx, ndy = ny + inv * sy, ndz = nz + inv * sz;
                    const float eps = 1e-8f;
                    if (fabsf(ndx) < eps && fabsf(ndy) < eps && fabsf(ndz) < eps) ndx = nx, ndy = ny, ndz = nz;
                    const bool odd = kind == MK_LAMBERT_CHECKER && checker_odd(px, py, pz);
                    at_r = odd ? q2.x : q1.x, at_g = odd ? q2.y : q1.y, at_b = odd ? q2.z : q1.z;
                } else if (kind == MK_METAL) {  // metal::scatter, material.h:47-53
                    const float inv = 1.0f / sqrtf(ra);
                    const float ux = inv * dx, uy = inv * dy, uz = inv * dz;
                    const float k2 = 2.0f * dot3(ux, uy, uz, nx, ny, nz);
                    const float rx = fmaf(-k2, nx, ux), ry = fmaf(-k2, ny, uy), rz = fmaf(-k2, nz, uz);
                    ndx = fmaf(q0.y, sx, rx), ndy = fmaf(q0.y, sy, ry), ndz = fmaf(q0.y, sz, rz);
                    at_r = q1.x, at_g = q1.y, at_b = q1.z;
                    scattered = dot3(ndx, ndy, ndz, nx, ny, nz) > 0.0f;
                } else if (kind == MK_DIELECTRIC) {  // dielectric::scatter, material.h:66-95
                    const float ratio = front ? q0.z : q0.y;
                    const float inv = 1.0f / sqrtf(ra);
                    const float ux = inv * dx, uy = inv * dy, uz = inv * dz;
                    const float udn = dot3(ux, uy, uz, nx, ny, nz);
                    const float cos_t = fminf(-udn, 1.0f);
                    const float sin_t = sqrtf(fmaf(-cos_t, cos_t, 1.0f));
                    bool refl = ratio * sin_t > 1.0f;
                    if (!refl) {
                        const float r0 = front ? q0.w : q1.w;
                        const float xx = 1.0f - cos_t;
                        const float x2 = xx * xx;
                        const float x5 = (x2 * x2) * xx;
                        refl = fmaf(1.0f - r0, x5, r0) > rng_next<COUNT>(rng, k0, k1);
                    }
                    if (refl) {  // reflect(), vec3.h:144-147
                        const float k2 = 2.0f * udn;
                        ndx = fmaf(-k2, nx, ux), ndy = fmaf(-k2, ny, uy), ndz = fmaf(-k2, nz, uz);
                    } else {  // refract(), vec3.h:149-155
                        const float ppx = ratio * fmaf(cos_t, nx, ux);
                        const float ppy = ratio * fmaf(cos_t, ny, uy);
                        const float ppz = ratio * fmaf(cos_t, nz, uz);
                        const float kk = -sqrtf(fabsf(1.0f - dot3(ppx, ppy, ppz, ppx, ppy, ppz)));
                        ndx = fmaf(kk, nx, ppx), ndy = fmaf(kk, ny, ppy), ndz = fmaf(kk, nz, ppz);
                    }
                    at_r = at_g = at_b = 1.0f;
                } else {  // diffuse_light: emitted, never scatters (material.cuh:161-182, main.cu:48-58)
                    const bool odd = kind == MK_LIGHT_CHECKER && checker_odd(px, py, pz);
                    const float er = odd ? q2.x : q1.x, eg = odd ? q2.y : q1.y, eb = odd ? q2.z : q1.z;
                    L_r = fmaf(er, beta_r, L_r), L_g = fmaf(eg, beta_g, L_g), L_b = fmaf(eb, beta_b, L_b);
                    scattered = false;
                    ndx = ndy = ndz = 0.0f;
                    at_r = at_g = at_b = 0.0f;
                }

                if (scattered) {
                    beta_r *= at_r, beta_g *= at_g, beta_b *= at_b;
                    ox = px, oy = py, oz = pz;
                    dx = ndx, dy = ndy, dz = ndz;
                    ra = dot3(dx, dy, dz, dx, dy, dz);
                    rinv_a = 1.0f / ra;
                    depth--;
                    path_done = depth <= 0;  // main.cpp:42 / main.cu:69
                } else {
                    path_done = true;  // absorbed: main.cpp:32 / main.cu:55-58
                }
            } else {
                // miss: main.cpp:36-38 (sky) or main.cu:63 (constant background)
                float bg_r, bg_g, bg_b;
                if (P.flags & RT_FLAG_SKY_GRADIENT) {
                    const float inv = 1.0f / sqrtf(ra);
                    const float t = 0.5f * (inv * dy + 1.0f);
                    const float omt = 1.0f - t;
                    bg_r = fmaf(t, 0.5f, omt), bg_g = fmaf(t, 0.7f, omt), bg_b = fmaf(t, 1.0f, omt);
                } else {
                    bg_r = P.background[0], bg_g = P.background[1], bg_b = P.background[2];
                }
                L_r = fmaf(beta_r, bg_r, L_r), L_g = fmaf(beta_g, bg_g, L_g), L_b = fmaf(beta_b, bg_b, L_b);
                path_done = true;
                if (COUNT) c_misses++;
            }
            if (path_done) {  // res += ray_color(...), main.cu:100
                sum_r += L_r, sum_g += L_g, sum_b += L_b;
                active = false;
            }
        }
    }

    // ---- coalesced framebuffer store: transpose the wave's 8x8x3 tile through LDS so
    // each store instruction writes the 96-byte row segments contiguously
    // (image[y*W + x] = res, main.cu:104; layout rgb_sum[(row*W + x)*3 + c])
    float *st = stage + wave * 192;
    st[lane * 3 + 0] = sum_r;
    st[lane * 3 + 1] = sum_g;
    st[lane * 3 + 2] = sum_b;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): LDS writes landed (wave-private region)
    const int x0 = bx * 32 + wave * 8;
    const int lr0 = band * 8;
    const size_t plane = (size_t)P.local_rows * P.width * 3;
    float *dst = out + (size_t)chunk * plane;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int f = lane + 64 * r;
        const int row = f / 24, col = f - row * 24;
        const int xx = x0 + col / 3;
        const int rr = lr0 + row;
        if (xx < P.width && rr < P.local_rows) dst[((size_t)rr * P.width + x0) * 3 + col] = st[f];
    }

    if (COUNT) {
        // one atomic per wave per counter
        auto wave_add = [&](unsigned long long *p, uint32_t v) {
            unsigned long long t = v;
            for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
            if (lane == 0 && t) atomicAdd(p, t);
        };
        wave_add(&counters->samples, c_samples);
        wave_add(&counters->queries, c_queries);
        wave_add(&counters->hits, c_hits);
        wave_add(&counters->misses, c_misses);
        wave_add(&counters->scatter[0], c_scatter0);
        wave_add(&counters->scatter[1], c_scatter1);
        wave_add(&counters->scatter[2], c_scatter2);
        wave_add(&counters->scatter[3], c_scatter3);
        wave_add(&counters->rng_draws, rng.draws);
    }
}

// chunk partial sums -> pixel sums, in chunk order (0 + c0 + c1 + ...)
__global__ __launch_bounds__(256) void reduce_chunks_kernel(const float *__restrict__ partial, float *__restrict__ out,
                                                            size_t plane, int num_chunks) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= plane) return;
    float t = 0.0f;
    for (int c = 0; c < num_chunks; ++c) t += partial[(size_t)c * plane + i];
    out[i] = t;
}

// launchers used by render_hip.cpp (host code compiled by the host compiler pass)
void launch_render(const RenderParams &P, const void *image, float *out, DevCounters *counters, size_t lds_bytes,
                   unsigned grid, hipStream_t stream) {
    if (counters)
        hipLaunchKernelGGL(render_kernel<true>, dim3(grid), dim3(256), lds_bytes, stream, P,
                           (const float4 *)image, out, counters);
    else
        hipLaunchKernelGGL(render_kernel<false>, dim3(grid), dim3(256), lds_bytes, stream, P,
                           (const float4 *)image, out, (DevCounters *)nullptr);
}

void launch_reduce(const float *partial, float *out, size_t plane, int num_chunks, hipStream_t stream) {
    unsigned grid = (unsigned)((plane + 255) / 256);
    hipLaunchKernelGGL(reduce_chunks_kernel, dim3(grid), dim3(256), 0, stream, partial, out, plane, num_chunks);
}

int set_max_dynamic_lds(size_t bytes) {
    hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void *>(&render_kernel<false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void *>(&render_kernel<true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    return (e1 == hipSuccess && e2 == hipSuccess) ? 0 : 1;
}

}  // namespace rtmi
