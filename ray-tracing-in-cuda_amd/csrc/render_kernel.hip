// The hot path: per-pixel path tracing on gfx950 (MI355X).
//
// Replaces   render<<<(W/8+1,H/8+1),(8,8)>>>   gpu-version/main.cu:72-105
//            ray_color                         gpu-version/main.cu:17-70  (semantics:
//                                              cmake-cpu-version/main.cpp:13-43)
//            hittable_list::hit + sphere/rect/cylinder::hit   gpu-version/object.cuh
//            material::scatter / emitted       gpu-version/material.cuh
//            camera::get_ray                   cmake-cpu-version/camera.h:32-39
//            curand XORWOW per-pixel state     -> Philox-seeded xorshift128 per (pixel, sample) (philox.h)
//
// Shape of the kernel
//   * 256-thread workgroup = 4 wave64; the grid only fills the chip (persistent waves) and
//     every wave pulls work items -- an 8x8 pixel tile x 64 sample indices -- from one
//     global counter.  A work-item (lane) owns ONE PATH at a time and runs its whole
//     bounce loop (get_color), exactly as the reference's thread does for its pixel.
//   * lanes stay converged across bounces: every iteration of the main loop is one
//     closest-hit query for all live lanes over a wave-uniform primitive loop.  A lane
//     whose path ended adds the radiance to its pixel and immediately takes the next
//     (pixel, sample) item of the tile's pool -- __ballot of the idle lanes, mbcnt for
//     the rank, a wave-uniform cursor -- so no lane waits for the slowest pixel of the
//     tile; the loop leaves on !__any(active) with the pool empty.
//     (variant bit 0 restores strict ownership: a lane only takes samples of its own
//     pixel.  Results are identical; it is kept to measure what the pool buys.)
//   * that hand-off is legal because the per-pixel sum is exact: each fp32 sample is
//     added as 64-bit fixed point (2^-24), in LDS per tile, then one 64-bit global
//     atomic per pixel and channel; a second kernel converts to the fp32 framebuffer
//     with fully coalesced stores.
//   * the primitive tables the inner loop reads ("hot" part of the scene image,
//     device_scene.h) are copied into LDS once per workgroup; all lanes read the same
//     record (broadcast ds_read_b128), four spheres are fetched one batch ahead of use.
//     No virtual calls, no pointer chasing; cold data (1/r, material records) stays in
//     global memory / L2 and is read once per bounce.
//   * the list is not scanned blindly: after the few big spheres, clusters of 8 spheres are
//     tested only when some lane's ray reaches the cluster's bounding box (aabb.hpp slab test
//     per lane + one __ballot), with a per-lane margin that covers the fp32 error of the
//     sphere test so that the closest hit -- and the framebuffer -- stay bit-identical to the
//     full scan (variant bit 4 is the full scan).
//   * arithmetic: fp32, every fused multiply-add explicit (-ffp-contract=off), IEEE
//     sqrt and divide, so results are bit-identical to the scalar restatement the
//     tests check against.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "device_scene.h"
#include "philox.h"
#include "rt_trig.h"
#include "../../include/rtmi.h"

// minimum resident waves per SIMD the register allocator must leave room for (8 <=> 64 VGPRs)
#ifndef RT_WAVES_PER_SIMD
#define RT_WAVES_PER_SIMD 7
#endif
// Wave priority (s_setprio) of the sections of an iteration.  Seven waves share a SIMD's issue port; a wave in the closest-hit
// query is a chain of short dependent steps (LDS reads, compares, branches) that wants its slot the moment its data is there,
// a wave in the seeding or in the rejection loop is a long run of independent vector instructions that can fill any gap.
// With every section at priority 0 the frame takes 131.5 ms; query + hit record + pixel accumulation at 2, scatter step /
// camera ray at 1, refill and rejection loop at 0: 127.2 ms (-3.3 %).  Measured: walk alone at 1 / 2 / 3: 129.7 / 130.1 /
// 129.5; query set-up + walk at 1: 128.6; + hit record and accumulation: 127.6; + scatter at 1: 127.4; the rejection loop
// or the refill raised instead: 131.6 / 128.8; the walk LOWERED: 133.4.  Re-measured on round 3's kernel (121.8 ms): the refill at
// 1 -- level with the scatter step -- 121.0 (-0.6 %, three alternations on one box), at 2: 122.2; the rejection loop at 1: 123.5;
// the walk and the hit record at 3 on top of that: 120.5 against 120.9 (means of five alternations; the walk alone at 3: 120.8, the
// query set-up at 3 as well: 121.05, at 1: 121.1, the scatter step at 2: 120.6).
#ifndef RT_PRIO_Q
#define RT_PRIO_Q 2  /* query set-up: prefix spheres, grid entry */
#endif
#ifndef RT_PRIO_W
#define RT_PRIO_W 3  /* grid walk */
#endif
#ifndef RT_PRIO_H
#define RT_PRIO_H 3  /* from the end of the walk to the refill: other primitives, hit record, pixel accumulation */
#endif
#ifndef RT_PRIO_F
#define RT_PRIO_F 1  /* refill (seeding, jitter) */
#endif
#ifndef RT_PRIO_R
#define RT_PRIO_R 0  /* rejection loop */
#endif
#ifndef RT_PRIO_S
#define RT_PRIO_S 1  /* after the rejection loop: scatter step, camera ray, ray tail */
#endif
// 1: the rejection loop draws three values per attempt for every lane and selects the state to keep (no inner branch)
// candidate predicate of a sphere test: a real root that is not behind the origin (as two nested branches: evaluated without
// short-circuit -- three compares, one branch -- it measured 146.5 against 145.5 ms)
#define RT_CAND(disc, hb, cc) (!((disc) < 0.0f) && !((hb) >= 0.0f && (cc) >= 0.0f))

namespace rtmi {

static constexpr float kTMin = 0.001f;  // main.cu:45 / main.cpp:22
static_assert(RT_FIX_BITS == RT_ACC_FIX_BITS, "the kernel's pixel sums and the ABI's scale");

// ---------------------------------------------------------------- RNG
struct LaneRng {
    Xor128 g;
    uint32_t draws;
};

__device__ __forceinline__ void rng_start(LaneRng &r, uint32_t pixel, uint32_t sample, uint32_t k0, uint32_t k1) {
    // The key schedule (k + r W for the ten rounds) is wave-uniform and loop-invariant, so the compiler computes the twenty
    // words once per launch -- and then, out of scalar registers, keeps them in the lanes of a spill VGPR and fetches
    // them with v_readlane (plus hazard nops) in every seeding.  Behind this barrier the key is a fresh scalar of the
    // iteration, and the schedule is two s_add per round.
    asm volatile("" : "+s"(k0), "+s"(k1));
    r.g = xor128_seed(pixel, sample, k0, k1);
}

template <bool COUNT>
__device__ __forceinline__ float rng_next(LaneRng &r) {
    const uint32_t w = xor128_next(r.g);
    if (COUNT) r.draws++;
    return (float)(w >> 8) * (1.0f / 16777216.0f);
}

template <bool COUNT>
__device__ __forceinline__ float rng_pm1(LaneRng &r) {  // random_double(-1, 1): -1 + 2 xi
    // xi = k 2^-24 with a 24-bit integer k: 2 xi and -1 + 2 xi are exact in fp32 (multiples of 2^-23 in [-1, 1)), so
    // one fused multiply-add returns the very value of the checker's three operations (convert, scale, shift)
    const uint32_t w = xor128_next(r.g);
    if (COUNT) r.draws++;
    return fmaf((float)(w >> 8), 1.0f / 8388608.0f, -1.0f);
}

__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) {
    return fmaf(ax, bx, fmaf(ay, by, az * bz));
}

// IEEE-correct fp32 square root, bit for bit what sqrtf() returns.  hipcc expands sqrtf() to v_sqrt_f32 (1 ulp) plus the
// one-step correction below, wrapped in a scaling for arguments below 2^-96 (v_sqrt_f32 flushes denormals) and a fix-up for
// 0 and inf: 17 instructions.  The arguments of this kernel (discriminants, squared lengths) are ordinary numbers, so the
// wrapping only runs -- through sqrtf() itself -- when some lane of the wave really holds such an argument: 11 instructions
// otherwise.  The kernel takes ~10 square roots per iteration of its main loop.
__device__ __forceinline__ float rt_sqrtf(float x) {
    // [2^-96, inf): one unsigned compare on the bit pattern (negative numbers and NaN fall outside as well)
    if (__builtin_expect((uint32_t)(__float_as_uint(x) - 0x0f800000u) >= (0x7f800000u - 0x0f800000u), 0)) return sqrtf(x);
    const float s = __builtin_amdgcn_sqrtf(x);
    const float s_dn = __uint_as_float(__float_as_uint(s) - 1u), s_up = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_dn = fmaf(-s_dn, s, x), r_up = fmaf(-s_up, s, x);
    float r = r_dn <= 0.0f ? s_dn : s;
    r = r_up > 0.0f ? s_up : r;
    return r;
}

// number of set bits of a lane mask as a 32-bit SCALAR (popcll's result is compared as a 64-bit value, for which the
// scalar unit has no ordered compare: the comparison then runs on the vector ALU, once per pass of the walk's loops)
__device__ __forceinline__ int mask_count(unsigned long long m) {
    return __builtin_amdgcn_readfirstlane(__builtin_popcount((uint32_t)m) + __builtin_popcount((uint32_t)(m >> 32)));
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

// a x b, one fused multiply-add per component (the checker uses the same form)
__device__ __forceinline__ void cross3(float ax, float ay, float az, float bx, float by, float bz, float &cx, float &cy, float &cz) {
    cx = fmaf(ay, bz, -(az * by));
    cy = fmaf(az, bx, -(ax * bz));
    cz = fmaf(ax, by, -(ay * bx));
}

// image texture, taichi-version/material.py:137-144: texel[int(frac(u) * rows)][int(frac(v) * cols)] / 255
__device__ __forceinline__ void image_texel(const float4 *__restrict__ image, const float4 q1, float u, float v, float &r, float &g,
                                            float &b) {
    const int rows = __float_as_int(q1.y), cols = __float_as_int(q1.z);
    const int x = min((int)((u - floorf(u)) * (float)rows), rows - 1);
    const int y = min((int)((v - floorf(v)) * (float)cols), cols - 1);
    const uint32_t w = reinterpret_cast<const uint32_t *>(image)[__float_as_int(q1.x) + x * cols + y];
    r = (float)(w & 255u) / 255.0f, g = (float)((w >> 8) & 255u) / 255.0f, b = (float)((w >> 16) & 255u) / 255.0f;
}

// original list index of a grouped primitive id (cold tables), for the tie rule
__device__ __forceinline__ int list_index_of(const RenderParams &P, const float4 *__restrict__ image, int id) {
    if (id < P.ns) return __float_as_int(image[P.off_sph_cold + id].z);
    if (id < P.ns + P.nr) return __float_as_int(image[P.off_rect_cold + (id - P.ns)].y);
    if (id < P.ns + P.nr + P.nc) return __float_as_int(image[P.off_cyl_cold + 4 * (id - P.ns - P.nr) + 3].y);
    return __float_as_int(image[P.off_tri_cold + 2 * (id - P.ns - P.nr - P.nc)].y);
}

// radiance sample -> 64-bit fixed point with RT_FIX_BITS (24) fractional bits, round to nearest even;
// NaN -> 0, magnitude clamped to RT_FIX_CLAMP (2^16).  |sample| <= 2^16 and at most 2^23 samples per pixel
// (checked by the host) keep every pixel sum below 2^39 < 2^63 / 2^24: the integer sums never wrap.
__device__ __forceinline__ unsigned long long radiance_to_fixed(float v) {
    // |v| < 128 (every sample that is not a look straight into a bright emitter): v * 2^24 is exact (a power of two)
    // and below 2^31, so round-to-nearest-even and a 32-bit convert give llrint((double)v * 2^24); sign-extended
    if (fabsf(v) < 128.0f) return (unsigned long long)(long long)(int)rintf(v * 16777216.0f);
    if (!(fabsf(v) <= RT_FIX_CLAMP)) v = (v != v) ? 0.0f : copysignf(RT_FIX_CLAMP, v);
    // the general case without fp64: |v| = hi + frac with hi = trunc(|v|) (v_cvt_u32_f32; the
    // subtraction of the integer part is exact), frac * 2^24 < 2^24 is exact in fp32 arithmetic before the
    // rounding, and rounding it to nearest even rounds the whole value to nearest even because hi * 2^24 is
    // an even integer.
    const float a = fabsf(v);
    const uint32_t hi = (uint32_t)a;
    const float frac = a - (float)hi;
    const uint32_t lo = (uint32_t)rintf(frac * 16777216.0f);
    const unsigned long long m = ((unsigned long long)hi << RT_FIX_BITS) + lo;
    return v < 0.0f ? 0ull - m : m;
}

// ---------------------------------------------------------------- kernel
// POOL:     idle lanes take the next (pixel, sample) item of the wave's tile (default)
//           / false: a lane only renders samples of its own pixel (ablation: the north_star's literal lane-per-pixel shape)
// SCALAR:   nothing is staged in LDS: every table is read from global memory -- wave-uniform reads through
//           the scalar cache (SGPR operands), per-lane reads through the vector L1/L2.  The mode for scenes
//           whose tables would crowd out occupancy or not fit the 160 KB at all.
// CULL:     the candidate search of the closest-hit query.  Every search is conservative with respect to the fp32 error of
//           the primitive tests, so the closest hit -- and the framebuffer -- equal the reference's linear scan bit for bit.
//           5  uniform grid over the clustered spheres, COMPACT tables (16-bit list entries, one word per cell): sphere-only
//              scenes whose tables fit LDS beside full occupancy.  Every lane walks the cells its ray crosses front to back.
//           6  the same for a grid one cell high (a sheet of spheres on the ground: RTIOW): the walk has no y axis
//           7  the same walk over the WIDE tables (32-bit list entries, two words per cell, up to 1023 cells per axis): every
//              other scene.  Its cells also list the rectangles, cylinders and triangles (taichi-version/bvh.py:109-199
//              indexes every hittable): a lane tests what its cells list; only the oversized primitives (the RTIOW ground,
//              a room's walls) are tested for every query
//           ablations (RTMI_ABLATIONS builds): 3 range tables, 2 two-level box hierarchy per lane (round 1's default), 1 wave
//           votes per cluster box (aabb.hpp:15-29 + __any), 0 no culling: the reference's linear hittable_list scan
// EXT:      the Taichi renderer's extras -- triangles (taichi-version/hittable.py:38-71) and image textures read at the
//           hit record's (u, v) (material.py:137-144).  Only the builds for scenes that use them carry the code: it
//           costs the kernel its register budget (86 spilled VGPRs instead of 23) whether a scene uses it or not.
// SPH:      the scene holds spheres only (RTIOW, the 3-sphere scene): no rectangle / cylinder / triangle code, no dispatch
//           on the winner's type -- and, what pays, a dozen fewer launch values and loop-invariant masks competing for scalar
//           registers (the surplus of those lives in the lanes of a spill VGPR: one v_readlane per use).  The compact-table
//           kernels (CULL 5, 6) are built this way only.
template <bool COUNT, bool POOL, bool SCALAR, int CULL, bool EXT, bool SPH>
__global__ __launch_bounds__(256, RT_WAVES_PER_SIMD) void render_kernel(const RenderParams P, const float4 *__restrict__ image,
                                                     unsigned long long *__restrict__ acc,
                                                     unsigned int *__restrict__ queue,
                                                     DevCounters *__restrict__ counters) {
    extern __shared__ float4 lds[];
    constexpr bool GRID = CULL == 5 || CULL == 6 || CULL == 7, SHEET = CULL == 6, WIDE = CULL == 7;
    constexpr int CSIZE = RT_CLUSTER;
    static_assert(!(CULL == 5 || CULL == 6) || SPH, "the compact grid tables list spheres only");
    static_assert(!(SPH && EXT), "image textures and triangles come with the general builds");
    // stage the hot tables (hittable_list contents) into LDS: each candidate search stages the part it reads
    // (the cluster searches leave the grid tables, which lie in front of their boxes, out: `gap` records)
    const int gap = (SCALAR || GRID) ? 0 : P.off_box - P.off_grid;
    const int staged = SCALAR ? 0 : (GRID ? P.hot_vec4_grid : (CULL == 3 ? P.hot_vec4_tables : P.hot_vec4) - gap);
    for (int i = threadIdx.x; i < staged; i += 256) lds[i] = image[i < P.off_grid ? i : i + gap];
    // per-wave tile accumulator of the current work item: 64 pixels x rgb, 64-bit fixed point
    unsigned long long *tile_acc = reinterpret_cast<unsigned long long *>(lds + staged);
    for (int i = threadIdx.x; i < 4 * 64 * 3; i += 256) tile_acc[i] = 0ull;
    __syncthreads();

    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;  // (wave: in a scalar register)
    unsigned long long *my_acc = tile_acc + wave * 192;
    const uint32_t k0 = P.seed_lo, k1 = P.seed_hi;
    const float4 *hot = SCALAR ? image : lds;
    const float4 *sph = hot;
    const float4 *rect = hot + P.off_rect_hot;
    const float4 *cyl = hot + P.off_cyl_hot;
    const int ns = P.ns, nr = SPH ? 0 : P.nr, nc = SPH ? 0 : P.nc, nt = SPH ? 0 : P.nt;
    const float4 *tri = hot + P.off_tri_hot;
    // original list index of a grouped primitive id (the tie rule)
    auto lidx = [&](int id) { return SPH ? __float_as_int(image[P.off_sph_cold + id].z) : list_index_of(P, image, id); };
    // u = (x + xi) / (W - 1), main.cu:96-97, evaluated as a multiply by the fp32 reciprocal (as the checker does)
    // (the two reciprocals come with the launch parameters: computed here they were vector registers, spilled to scratch and
    //  fetched back with two dependent scratch loads in every refill)

    uint32_t c_samples = 0, c_queries = 0, c_hits = 0, c_misses = 0;
    uint32_t c_scatter0 = 0, c_scatter1 = 0, c_scatter2 = 0, c_scatter3 = 0;
    uint32_t c_cand = 0, c_cand_wave = 0, c_clusters = 0, c_groups = 0, c_wave_queries = 0, c_lane_clusters = 0, c_lane_groups = 0, c_lane_cands = 0, c_group_maxpop = 0, c_query_maxpop = 0;
    // COUNT: shader-clock time of the main loop's sections, per wave (refill, prefix spheres, culled spheres +
    // rects + cylinders, shading, pixel accumulation, loop control)
    unsigned long long cyc[6] = {0, 0, 0, 0, 0, 0}, tmark = 0, t_qe = 0;
    auto tick = [&](int section) {
        if (COUNT) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            cyc[section] += now - tmark;
            tmark = now;
        }
    };
    if (COUNT) tmark = __builtin_amdgcn_s_memtime();
    unsigned long long t_begin = 0, c_begin = 0;
    if (COUNT) t_begin = __builtin_amdgcn_s_memrealtime(), c_begin = __builtin_amdgcn_s_memtime();
    if (COUNT && lane == 0) {
        const unsigned long long t = t_begin;
        atomicMin(&counters->t_start_min, t);
        atomicMax(&counters->t_start_max, t);
    }

    // ---- persistent waves, streaming work items.  The grid only fills the chip; every wave pulls
    // (8x8 tile, sample chunk) work items from one global counter until it runs dry.  A wave does not
    // drain an item before taking the next: when the pool of the current item is handed out and a lane
    // is idle, the item's tile accumulator (LDS) is flushed and the paths still alive finish as ORPHANS
    // that add their sample straight to the global accumulators (integer sums: any split of an item's
    // additions gives the same total).  So lanes only idle at the very end of the launch, and items can
    // be short.  From here on the four waves of the workgroup never synchronise again.
    unsigned long long *c_acc = my_acc;  // accumulator of the current item (wave-uniform pointer)
    int c_x0 = 0, c_band = 0, c_sbegin = 0, c_pool = 0, cursor = 0;  // c_pool = 64 x samples of the item
    bool c_valid = false, queue_empty = false;
    int c_hy = 0, c_hvalid = 0;  // this lane's home pixel in the current item (row, on-image)
    unsigned long long c_hvmask = 0ull;  // ... and the on-image bits of all 64 home pixels (wave-uniform)
    int mine = 0;                // !POOL: samples of the home pixel started so far (current item)

    // home pixel of this lane in the tile (x0, band): column, dense local row, image row, on-image
    // item-level launch values, read where they are needed (device_scene.h, ItemParams): wave-uniform
    // 16-byte loads behind a compiler barrier, so that they are neither hoisted out of the main loop (and
    // then spilled) nor kept in registers between items
    auto ipar4 = [&](int quad) {
        asm volatile("" ::: "memory");
        return reinterpret_cast<const int4 *>(queue + RT_ITEM_PARAMS_AT)[quad];
    };
    auto home_pixel = [&](int x0, int band, int &hx, int &hlr, int &hy, int &hvalid) {
        const int4 g = ipar4(3);  // {tile_rows, tile_first, tile_stride, local_rows}
        const int tile_rows = g.x, tile_first = g.y, tile_stride = g.z, local_rows = g.w;
        const int tile_rotate = ipar4(2).z;
        hx = x0 + (lane & 7);
        hlr = band * 8 + (lane >> 3);
        const int htl = hlr / tile_rows;
        int tile = tile_first + htl * tile_stride;
        if (tile_rotate == 2) {  // there and back (include/rtmi.h, rt_opts.tile_rotate)
            tile = (htl >> 1) * 2 * tile_stride + ((htl & 1) ? 2 * tile_stride - 1 - tile_first : tile_first);
        } else if (tile_rotate) {  // rotated interleave
            int j = (tile_first - htl) % tile_stride;
            if (j < 0) j += tile_stride;
            tile = htl * tile_stride + j;
        }
        hy = tile * tile_rows + (hlr - htl * tile_rows);
        hvalid = (hx < P.width && hlr < local_rows && hy < P.height) ? 1 : 0;
    };
    // tile accumulator -> global accumulators (image[y*W + x] += res, main.cu:104; other sample chunks of the same
    // pixels are other work items, hence atomics), then clear it for reuse.  The tile's 192 sums lie in LDS as
    // [row][column][channel], which is also the order of a tile row in the global plane (24 consecutive 64-bit
    // words): lane l adds the words l, 64 + l and 128 + l, so one instruction covers 512 contiguous bytes of LDS and
    // 2 2/3 tile rows of 192 contiguous bytes in memory (the atomics execute memory-side in 64-byte requests: with
    // one pixel per lane -- 24 bytes apart -- every request carried 8 useful bytes).  Words that are zero are
    // skipped: black samples, and the pixels of a ragged tile outside the image or the shard, which never receive a
    // sample -- so no bounds logic is needed here.
    auto flush_tile = [&](unsigned long long *tile, int x0, int band) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int k = j * 64 + lane;
            const unsigned long long v = tile[k];
            if (v != 0ull) {
                const int row = k / 24, rem = k - row * 24;
                atomicAdd(acc + ((size_t)(band * 8 + row) * P.width + x0) * 3 + rem, v);
                tile[k] = 0ull;
            }
        }
        __builtin_amdgcn_wave_barrier();
    };

    LaneRng rng;
    rng.g.x = rng.g.y = rng.g.z = 0u, rng.g.w = 1u, rng.draws = 0;

    float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 1, ra = 1, rinv_a = 1;
    // (no radiance accumulator: in this integrator a path collects radiance exactly once, in the event that ends it -- the sky
    //  or the background on a miss, main.cpp:36-38 / main.cu:63, or an emitter, which never scatters, main.cu:48-58 -- so
    //  ray_color's accumulated colour is throughput x that event's radiance, and a path that is absorbed, runs out of depth
    //  or loses the roulette contributes exactly zero: nothing to add)
    float beta_r = 1, beta_g = 1, beta_b = 1;
    int depth = 0;
    // where this lane's live path adds its sample: >= 0 the tile-local pixel (0 .. 63) in the current item's accumulator; < 0 the
    // path outlived its item (an orphan): ~cur_p is its dense local pixel index for a direct global add
    int cur_p = lane;
    bool active = false;

    // One iteration of the main loop:
    //   (1) closest-hit query of every live lane              hittable_list::hit
    //   (2) hit record + material of the winner; a miss or an emitter ends the path here
    //   (3) pixel accumulation of the paths that ended
    //   (4) refill: idle lanes take the next (pixel, sample) of the tile's pool, seed their stream, draw the jitter
    //   (5) ONE rejection loop for both kinds of lanes: random_in_unit_sphere for the scatter step of a lambertian /
    //       metal hit (three draws per attempt) and random_in_unit_disk for the lens sample of a new path (two)
    //   (6) the scatter step (lambertian / metal / dielectric) | the camera ray, then what both share: |d|^2, 1 / |d|^2
    // The loops of (5) cost max-over-lanes attempts each; as two loops (one inside the refill, one inside the shading)
    // they took 13 % of the frame (measured by cutting them out).
    // CULL == 5: ray parameter at which this lane's grid walk was cut short in the previous iteration (0: it was not);
    // the walk goes on from there in this one
    float t_res = 0.0f;
    for (;;) {
        tick(5);
        bool path_done = false;
        float L_r = 0, L_g = 0, L_b = 0;  // the sample of a path that ends in this iteration
        asm volatile("" : "=v"(L_r), "=v"(L_g), "=v"(L_b));  // (read by the lanes with path_done, which set them)
        // the winner's hit record, kept for the scatter step
        float px = 0, py = 0, pz = 0, nx = 0, ny = 0, nz = 0, inv_len = 0;
        int mat = 0, kind = -1;  // kind >= 0: a scatter step is due in (6)
        // (only lanes with kind >= 0 read the record, and they have written it; an "undefined" value from an empty asm
        //  spares the eight v_mov ..., 0 per iteration that the zero initialisers above cost)
        asm volatile("" : "=v"(px), "=v"(py), "=v"(pz), "=v"(nx), "=v"(ny), "=v"(nz), "=v"(inv_len), "=v"(mat));
        bool front = false;
        float tex_r = 0, tex_g = 0, tex_b = 0;  // the texel of an image texture at the hit's (u, v)
        if (__any(active)) {
        {
            // ---- closest-hit query over the LDS-resident list (hittable_list::hit,
            // object.cuh:23-37).  Wave-uniform trip counts; `best_id` is the grouped id.
            float best_t = INFINITY;
            int best_id = -1;

            // spheres: sphere::hit, object.cuh:47-75.  Early-outs that need no sqrt:
            //   disc < 0                      -> no real root
            //   hb >= 0 and c >= 0            -> both roots <= 0 < t_min  (then
            //     sqrt(disc) <= sqrt(fl(hb*hb)) = hb, so (-hb + sqrtd) <= 0 exactly)
            // The closest hit does not depend on the visiting order (ties go to the later
            // list entry, resolved through list_index_of), so the device table is sorted
            // big-spheres-first.
            auto resolve = [&](int idx, float hb, float disc) {
                if (COUNT) {  // diagnostic: candidate lanes, and entries of this block per wave
                    c_cand++;
                    const unsigned long long em = __builtin_amdgcn_ballot_w64(true);
                    if ((int)__builtin_ctzll(em) == lane) c_cand_wave++;
                }
                const float sq = rt_sqrtf(disc);
                float root = (-hb - sq) * rinv_a;
                if (root < kTMin || best_t < root) root = (-hb + sq) * rinv_a;
                if (!(root < kTMin || best_t < root)) {
                    bool take = true;
                    if (root == best_t && best_id >= 0)
                        take = lidx(idx) > lidx(best_id);
                    if (take) {
                        best_t = root;
                        best_id = idx;
                    }
                }
            };
            // the point where the ray meets a triangle's plane and its ray parameter (hittable.py:44-52, 61):
            // n = the unit normal turned towards the origin, theta = d.n / |d| < 0, r = o - d/|d| (oc.n) / theta
            // (|d| and d / |d| belong to the ray, not to the triangle: one square root and three divisions per query instead of per
            //  test -- a triangle test is about 190 instructions, 41 of them these)
            float tq_a = 1.0f, tq_ux = 0.0f, tq_uy = 0.0f, tq_uz = 0.0f;
            if (EXT && nt > 0 && active) {
                tq_a = rt_sqrtf(ra);
                tq_ux = dx / tq_a, tq_uy = dy / tq_a, tq_uz = dz / tq_a;
            }
            auto tri_plane = [&](const float4 r0, const float4 r1, const float4 r2, float &rix, float &riy, float &riz,
                                 float &root) -> bool {
                float tnx = r0.w, tny = r1.w, tnz = r2.w;
                float ocn = dot3(ox - r0.x, oy - r0.y, oz - r0.z, tnx, tny, tnz);
                if (ocn < 0.0f) tnx = -tnx, tny = -tny, tnz = -tnz, ocn = -ocn;
                const float theta = dot3(dx, dy, dz, tnx, tny, tnz) / tq_a;
                if (!(theta < 0.0f)) return false;
                rix = ox - (tq_ux * ocn) / theta;
                riy = oy - (tq_uy * ocn) / theta;
                riz = oz - (tq_uz * ocn) / theta;
                root = ((-ocn) / theta) / tq_a;
                return true;
            };
#define RT_SPHERE_TEST(S, IDX)                                                                 \
    {                                                                                          \
        const float ocx = ox - S.x, ocy = oy - S.y, ocz = oz - S.z;                            \
        const float hb = dot3(ocx, ocy, ocz, dx, dy, dz);                                      \
        const float cc = fmaf(ocx, ocx, fmaf(ocy, ocy, fmaf(ocz, ocz, -S.w)));                 \
        const float disc = fmaf(hb, hb, -(ra * cc));                                           \
        const bool cand = RT_CAND(disc, hb, cc);                                                \
        if (__builtin_expect(cand, 0)) resolve(IDX, hb, disc);                                 \
    }
            // the same test with the candidate PARKED in (p_idx, p_hb, p_disc) instead of resolved on the spot: the cold block
            // (IEEE sqrt, both roots, range and tie rules: ~45 instructions) runs once for the candidates of several
            // records.  The closest hit does not depend on the order in which candidates are resolved (a candidate's root
            // is its smallest one >= t_min, accepted while it is <= best_t: the outcome is the minimum over candidates, ties
            // by list index), so parking is free to reorder.  A lane that finds a second candidate resolves the first there.
#define RT_SPHERE_PARK(S, IDX)                                                                 \
    {                                                                                          \
        const float ocx = ox - S.x, ocy = oy - S.y, ocz = oz - S.z;                            \
        const float hb = dot3(ocx, ocy, ocz, dx, dy, dz);                                      \
        const float cc = fmaf(ocx, ocx, fmaf(ocy, ocy, fmaf(ocz, ocz, -S.w)));                 \
        const float disc = fmaf(hb, hb, -(ra * cc));                                           \
        const bool cand = RT_CAND(disc, hb, cc);                                                \
        if (__builtin_expect(cand, 0)) {                                                       \
            if (p_idx >= 0) resolve(p_idx, p_hb, p_disc);                                      \
            p_idx = IDX, p_hb = hb, p_disc = disc;                                             \
        }                                                                                      \
    }
            if (RT_PRIO_Q != RT_PRIO_S) __builtin_amdgcn_s_setprio(RT_PRIO_Q);
            if (active) {
            // the always-tested prefix (big spheres, largest first), four records at a time: the first one -- in RTIOW the
            // ground, a candidate for half of the lanes -- is resolved at once, the other three share one resolve
            for (int i = 0; i < P.np; i += 4) {
                const float4 s0 = sph[i], s1 = sph[i + 1], s2 = sph[i + 2], s3 = sph[i + 3];
                RT_SPHERE_TEST(s0, i)
                int p_idx = -1;
                float p_hb = 0.0f, p_disc = 0.0f;
                asm volatile("" : "=v"(p_hb), "=v"(p_disc));  // (read only where p_idx >= 0, which comes with their values)
                RT_SPHERE_PARK(s1, i + 1)
                RT_SPHERE_PARK(s2, i + 2)
                RT_SPHERE_PARK(s3, i + 3)
                if (p_idx >= 0) resolve(p_idx, p_hb, p_disc);
            }
            if (!CULL) {
                // flat scan (the reference's hittable_list loop) of every cluster's 8 records (clusters are 9 slots apart,
                // see the packer): 16 records per iteration -- two clusters, the never-hit slot between them skipped -- in two
                // register sets of four fetched half a step ahead of their use; the table ends with all-padding clusters, so
                // the last read-ahead stays inside it
                {
                    constexpr int kStep = 2 * (CSIZE + 1);
#define RT_OFF(k) ((k) + ((k) >= CSIZE ? 1 : 0))
                    const int iters = (P.ncl * CSIZE + 15) / 16;
                    int base = P.np;
                    float4 a0 = sph[base], a1 = sph[base + 1], a2 = sph[base + 2], a3 = sph[base + 3];
                    for (int it = 0; it < iters; ++it) {
#pragma unroll
                        for (int k = 0; k < 16; k += 8) {
                            const float4 b0 = sph[base + RT_OFF(k + 4)], b1 = sph[base + RT_OFF(k + 5)],
                                         b2 = sph[base + RT_OFF(k + 6)], b3 = sph[base + RT_OFF(k + 7)];
                            RT_SPHERE_TEST(a0, base + RT_OFF(k))
                            RT_SPHERE_TEST(a1, base + RT_OFF(k + 1))
                            RT_SPHERE_TEST(a2, base + RT_OFF(k + 2))
                            RT_SPHERE_TEST(a3, base + RT_OFF(k + 3))
                            const int nxt = k + 8 < 16 ? RT_OFF(k + 8) : kStep;
                            a0 = sph[base + nxt], a1 = sph[base + nxt + 1], a2 = sph[base + nxt + 2], a3 = sph[base + nxt + 3];
                            RT_SPHERE_TEST(b0, base + RT_OFF(k + 4))
                            RT_SPHERE_TEST(b1, base + RT_OFF(k + 5))
                            RT_SPHERE_TEST(b2, base + RT_OFF(k + 6))
                            RT_SPHERE_TEST(b3, base + RT_OFF(k + 7))
                        }
                        base += kStep;
                    }
#undef RT_OFF
                }
            }
            }
            tick(1);
            // ---- culling set-up (CULL): aabb::hit (aabb.hpp:15-29) for every live lane, then one wave-wide
            // vote per box; used for the sphere clusters and for each cylinder's bounding box
            // 1-ulp reciprocals are enough here: the box test only has to be conservative, and the
            // margin below is five orders of magnitude larger than their error
            // Clamped to +-1e18: a direction component that is exactly 0 (a fuzz-free mirror produces them)
            // would give inf, and the fma form below inf - inf = NaN on the face behind the origin, which
            // min/max then drop together with the slab.  With a huge finite value the axis keeps its
            // meaning: origin inside the slab -> (-huge, +huge), outside -> both of one sign -> dead.
            // The values are only needed around the box tests; they are derived where those sit (per window of
            // clusters, and once more for the cylinders' and triangles' boxes) instead of once per query, so that
            // the ten registers are free while the wave walks its clusters -- the kernel's register peak.
            struct BoxP {
                float idx, idy, idz, nxm, nym, nzm, nxp, nyp, nzp, marg;
            };
            auto box_params = [&]() -> BoxP {
                BoxP b;
                b.idx = __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(dx), -1e18f, 1e18f);
                b.idy = __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(dy), -1e18f, 1e18f);
                b.idz = __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(dz), -1e18f, 1e18f);
                // per-lane box margin covering the fp32 error of the sphere test at this origin's
                // distance (derivation in render_host.hip): two shifted origins, nothing per box
                b.marg = 4e-3f * (fmaxf(fmaxf(fabsf(ox), fabsf(oy)), fabsf(oz)) + P.cull_extent1);
                // slab distances as one fma per face: t = b * (1/d) - (o +- marg) * (1/d).  The products
                // cancel to an absolute error ~ eps |o/d|, i.e. ~1e-7 |o| in space: four orders of
                // magnitude inside the margin.
                b.nxm = -(ox + b.marg) * b.idx, b.nym = -(oy + b.marg) * b.idy, b.nzm = -(oz + b.marg) * b.idz;
                b.nxp = -(ox - b.marg) * b.idx, b.nyp = -(oy - b.marg) * b.idy, b.nzp = -(oz - b.marg) * b.idz;
                return b;
            };
            const float4 *box = hot + (P.off_box - gap);
            const float4 *gbox = hot + (P.off_gbox - gap);
            // best_t (1 + 1e-4), refreshed whenever spheres have been tested (a stale, larger value only
            // culls less)
            float blim = best_t * 1.0001f;
            auto slab_live = [&](const BoxP &b, const float4 bmn, const float4 bmx) -> bool {
                const float lx = fmaf(bmn.x, b.idx, b.nxm), ux = fmaf(bmx.x, b.idx, b.nxp);
                const float ly = fmaf(bmn.y, b.idy, b.nym), uy = fmaf(bmx.y, b.idy, b.nyp);
                const float lz = fmaf(bmn.z, b.idz, b.nzm), uz = fmaf(bmx.z, b.idz, b.nzp);
                // live  <=>  tn <= tf, tf >= 0, tn <= best_t (1 + 1e-4)
                //       <=>  max(tn, 0) <= min(tf, best_t (1 + 1e-4))          (NaN -> live)
                const float tn = fmaxf(fmaxf(fmaxf(fminf(lx, ux), fminf(ly, uy)), 0.0f), fminf(lz, uz));
                const float tf = fminf(fminf(fminf(fmaxf(lx, ux), fmaxf(ly, uy)), blim), fmaxf(lz, uz));
                return !(tn > tf);
            };
            // ---- the other primitives' tests (one primitive, index uniform or per lane)
            // axis-aligned rects: xy_rect/xz_rect/yz_rect::hit, object.cuh:105-192
            auto test_rect = [&](int j) {
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
            };
            // cylinders: cylinder::hit + quadratic, object.cuh:199-214, 233-290
            auto test_cyl = [&](int k) {
                const float4 r0 = cyl[RT_CYL_STRIDE * k], r1 = cyl[RT_CYL_STRIDE * k + 1], r2 = cyl[RT_CYL_STRIDE * k + 2], pr = cyl[RT_CYL_STRIDE * k + 3];
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
                    const float sq = rt_sqrtf(delta);
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
            };
            // triangles: hit_triangle, taichi-version/hittable.py:38-71 -- the plane of the triangle (its unit normal
            // turned towards the ray origin), then four same-side tests of the plane point
            auto test_tri = [&](int k) {
                const float4 r0 = tri[RT_TRI_STRIDE * k], r1 = tri[RT_TRI_STRIDE * k + 1], r2 = tri[RT_TRI_STRIDE * k + 2];
                float rix, riy, riz, root;
                if (tri_plane(r0, r1, r2, rix, riy, riz, root) && !(root < kTMin || root > best_t)) {
                    const float e21x = r1.x - r0.x, e21y = r1.y - r0.y, e21z = r1.z - r0.z;
                    const float e31x = r2.x - r0.x, e31y = r2.y - r0.y, e31z = r2.z - r0.z;
                    const float e32x = r2.x - r1.x, e32y = r2.y - r1.y, e32z = r2.z - r1.z;
                    const float a1x = rix - r0.x, a1y = riy - r0.y, a1z = riz - r0.z;
                    const float a2x = rix - r1.x, a2y = riy - r1.y, a2z = riz - r1.z;
                    float px_, py_, pz_, qx_, qy_, qz_;
                    cross3(a1x, a1y, a1z, e21x, e21y, e21z, px_, py_, pz_);
                    cross3(e31x, e31y, e31z, e21x, e21y, e21z, qx_, qy_, qz_);
                    const float n1 = dot3(px_, py_, pz_, qx_, qy_, qz_);
                    cross3(a2x, a2y, a2z, -e21x, -e21y, -e21z, px_, py_, pz_);
                    cross3(e32x, e32y, e32z, -e21x, -e21y, -e21z, qx_, qy_, qz_);
                    const float n2 = dot3(px_, py_, pz_, qx_, qy_, qz_);
                    cross3(a1x, a1y, a1z, e31x, e31y, e31z, px_, py_, pz_);
                    cross3(e21x, e21y, e21z, e31x, e31y, e31z, qx_, qy_, qz_);
                    const float n3 = dot3(px_, py_, pz_, qx_, qy_, qz_);
                    cross3(a2x, a2y, a2z, e32x, e32y, e32z, px_, py_, pz_);
                    cross3(-e21x, -e21y, -e21z, e32x, e32y, e32z, qx_, qy_, qz_);
                    const float n4 = dot3(px_, py_, pz_, qx_, qy_, qz_);
                    if (n1 > 0.0f && n2 > 0.0f && n3 > 0.0f && n4 > 0.0f) {
                        bool take = true;
                        if (root == best_t && best_id >= 0)
                            take = list_index_of(P, image, ns + nr + nc + k) > list_index_of(P, image, best_id);
                        if (take) {
                            best_t = root;
                            best_id = ns + nr + nc + k;
                        }
                    }
                }
            };
            bool far_scan = false;  // GRID: this lane's origin lies beyond the reach of the cells' lists: it scans what they list
            if (active) {
            if (GRID && P.grid_cells != 0) {  // (no cells: a scene small enough for every primitive to be tested per query)
                // ---- uniform grid, 3-D DDA per lane (the default).  The clustered spheres -- and, in the wide tables, the
                // rectangles, cylinders and triangles that are not oversized -- are listed in the cells their (error-grown, see
                // the packer) boxes touch; a lane walks the cells its ray crosses in the order it crosses them and tests what
                // they list, so the nearest hit ends the walk: a cell is only entered while its entry distance is within
                // best_t (1 + 1e-4).
                const float4 *gh = hot + P.off_grid;
                const float4 g_min = gh[0], g_inv = gh[1], g_size = gh[2];
                const int gnx = __float_as_int(gh[3].x), gny = __float_as_int(gh[3].y), gnz = __float_as_int(gh[3].z);
                const uint32_t *g_cells = reinterpret_cast<const uint32_t *>(hot + P.off_grid_cells);
                const uint16_t *g_items = reinterpret_cast<const uint16_t *>(hot + P.off_grid_items);
                const BoxP bp = box_params();
                // which tier of the cells' lists covers this lane's origin (the packer: |o| against ob_near, ob_far)
                const float o2 = fmaf(ox, ox, fmaf(oy, oy, oz * oz));
                const bool tier_far = o2 > g_min.w, beyond = o2 > g_inv.w;
                // cell header, compact: (first << 12) | (n_near << 6) | n_all;  wide: {first, n_near | n_all << 10 | n_other << 20}:
                // the sphere entries [first, + n_near) serve near origins, [first, + n_all) far ones, and the n_other entries
                // behind them are the cell's other primitives (grouped ids)
                const int cnt_shift = tier_far ? (WIDE ? 10 : 0) : (WIDE ? 0 : 6);
                constexpr int REM_BITS = WIDE ? 10 : 8;            // steps left per axis, packed in one register
                constexpr uint32_t REM_MASK = (1u << REM_BITS) - 1u, CNT_MASK = WIDE ? 1023u : 63u;
                constexpr bool OTHERS = WIDE && !SPH;
                const uint32_t *g_items32 = reinterpret_cast<const uint32_t *>(g_items);
                int ko = 0, koend = 0;  // OTHERS: the entries of the cell's other primitives still to test
                auto cell_list = [&](int cell, int &first, int &end) {  // a cell's list entries [first, end) of this lane's tier
                    if (WIDE) {
                        const uint2 h = reinterpret_cast<const uint2 *>(g_cells)[cell];
                        first = (int)h.x, end = first + (int)((h.y >> cnt_shift) & CNT_MASK);
                        if (OTHERS) ko = (int)h.x + (int)((h.y >> 10) & CNT_MASK), koend = ko + (int)(h.y >> 20);
                    } else {
                        const uint32_t h = g_cells[cell];
                        first = (int)(h >> 12), end = first + (int)((h >> cnt_shift) & CNT_MASK);
                    }
                };
                // the grid's bounds (un-grown: the lists carry the growth); the near tier's lie g_size.w further in
                const float shrink = tier_far ? 0.0f : g_size.w;
                const float bx0 = g_min.x + shrink, by0 = g_min.y + shrink, bz0 = g_min.z + shrink;
                const float bx1 = fmaf((float)gnx, g_size.x, g_min.x) - shrink, by1 = fmaf((float)gny, g_size.y, g_min.y) - shrink,
                            bz1 = fmaf((float)gnz, g_size.z, g_min.z) - shrink;
                blim = best_t * 1.0001f;
                // a walk that was cut short goes on a few ulps past the cell boundary it stopped at: inside the next cell (the
                // lists' margin of 0.004 cell covers the sliver), so that every resumption ends at a later boundary
                const float t_from = t_res * 1.000002f;
                bool live = false;
                int ci = 0, k = 0, kend = 0;
                uint32_t rem = 0;  // steps left before the ray leaves the grid: x | y << 8 | z << 16
                float tmx = INFINITY, tmy = INFINITY, tmz = INFINITY, t_exit = 0.0f;
                if (beyond) {
                    const float4 fmn = {bx0, by0, bz0, 0.0f}, fmx = {bx1, by1, bz1, 0.0f};
                    far_scan = slab_live(bp, fmn, fmx);
                } else {
                    // exact slab distances here (no margin): t = (b - o) * (1 / d), reciprocals clamped as in box_params
                    const float lx = (bx0 - ox) * bp.idx, ux = (bx1 - ox) * bp.idx;
                    const float ly = (by0 - oy) * bp.idy, uy = (by1 - oy) * bp.idy;
                    const float lz = (bz0 - oz) * bp.idz, uz = (bz1 - oz) * bp.idz;
                    const float tn = fmaxf(fmaxf(fmaxf(fminf(lx, ux), fminf(ly, uy)), t_from), fminf(lz, uz));
                    t_exit = fminf(fminf(fmaxf(lx, ux), fmaxf(ly, uy)), fmaxf(lz, uz));
                    live = !(tn > fminf(t_exit, blim));
                    if (live) {
                        // the cell of the entry point
                        const float px = fmaf(tn, dx, ox), py = fmaf(tn, dy, oy), pz = fmaf(tn, dz, oz);
                        const int ix = min(max((int)floorf((px - g_min.x) * g_inv.x), 0), gnx - 1);
                        // (SHEET: one layer of cells.  A ray that leaves it through the top or the bottom ends its walk at t_exit,
                        // the exit from the grid's bounds, which comes no later than the layer's own faces; so the walk needs
                        // neither a y cell index nor a y leave distance, and visits the cells the 3-D walk would visit)
                        const int iy = SHEET ? 0 : min(max((int)floorf((py - g_min.y) * g_inv.y), 0), gny - 1);
                        const int iz = min(max((int)floorf((pz - g_min.z) * g_inv.z), 0), gnz - 1);
                        ci = SHEET ? iz * gnx + ix : (iz * gny + iy) * gnx + ix;
                        // ray parameter at which the ray leaves the cell along each axis (a component of exactly 0
                        // never leaves), and how many steps are left before it leaves the grid
                        tmx = dx == 0.0f ? INFINITY : (fmaf((float)(ix + (dx > 0.0f ? 1 : 0)), g_size.x, g_min.x) - ox) * bp.idx;
                        if (!SHEET) tmy = dy == 0.0f ? INFINITY : (fmaf((float)(iy + (dy > 0.0f ? 1 : 0)), g_size.y, g_min.y) - oy) * bp.idy;
                        tmz = dz == 0.0f ? INFINITY : (fmaf((float)(iz + (dz > 0.0f ? 1 : 0)), g_size.z, g_min.z) - oz) * bp.idz;
                        rem = (uint32_t)(dx > 0.0f ? gnx - 1 - ix : ix) | (SHEET ? 0u : (uint32_t)(dy > 0.0f ? gny - 1 - iy : iy) << REM_BITS) |
                              (uint32_t)(dz > 0.0f ? gnz - 1 - iz : iz) << (2 * REM_BITS);
                        cell_list(ci, k, kend);
                        if (COUNT && t_res == 0.0f) c_lane_groups++, c_group_maxpop += tier_far ? 1u : 0u;
                    }
                }
                if (COUNT && far_scan) c_query_maxpop++;
                t_res = 0.0f;
                // |size / d| per axis: what one step adds to the leave distance
                const float dtx = g_size.x * fabsf(bp.idx), dty = g_size.y * fabsf(bp.idy), dtz = g_size.z * fabsf(bp.idz);
                const int sx = dx > 0.0f ? 1 : -1, sy = dy > 0.0f ? gnx : -gnx, sz = SHEET ? (dz > 0.0f ? gnx : -gnx) : (dz > 0.0f ? gnx * gny : -(gnx * gny));
                // cell by cell: the wave first drains the lists of the cells its lanes stand in (one sphere per lane and
                // pass), then every lane steps (measured: 47.4 ms against 54.8 for one flattened loop in which a lane either
                // tests or steps, RTIOW 256 spp)
#ifndef RT_STEP_AT
#define RT_STEP_AT 16  /* lanes that must be waiting before the wave runs a step pass while others still test (65: never; 16 / 24 / 32 / 65: 145.4 / 146.0 / 145.8 / 147.9 ms) */
#endif
#ifndef RT_WALK_TAIL
#define RT_WALK_TAIL 20     /* at most this many lanes still walking ...            (0: never cut; 0 / 8 / 12 / 20: 149.0 / 146.5 / 146.0 / 145.0 ms) */
#define RT_WALK_WAITING 32  /* ... and at least this many live lanes done: the stragglers go on next iteration */
#endif
                if (RT_PRIO_W != RT_PRIO_Q) __builtin_amdgcn_s_setprio(RT_PRIO_W);
                while (__builtin_amdgcn_ballot_w64(live) != 0ull) {
                    while (__builtin_amdgcn_ballot_w64(k < kend) != 0ull) {
                        if (COUNT) c_clusters++;
                        // two list entries per pass: both index reads, then both record reads, are in flight together, and a
                        // cell's list costs the wave ceil(n / 2) passes (each a dependent LDS round trip, exec-mask
                        // bookkeeping and a taken branch) instead of n.  A list of odd length reads the never-hit slot
                        // behind the first cluster for its second half.
                        if (k < kend) {
                            const int idx = WIDE ? (int)g_items32[k] : (int)g_items[k];
                            const int j_raw = WIDE ? (int)g_items32[k + 1] : (int)g_items[k + 1];  // (one entry past the list at worst: the next list, or the padding entry behind the last one)
                            const int jdx = k + 1 < kend ? j_raw : P.np + CSIZE;
                            if (COUNT) c_lane_clusters += k + 1 < kend ? 2u : 1u;
                            k += 2;
                            const float4 S = sph[idx], T = sph[jdx];
                            int p_idx = -1;
                            float p_hb = 0.0f, p_disc = 0.0f;
                            asm volatile("" : "=v"(p_hb), "=v"(p_disc));
                            RT_SPHERE_PARK(S, idx)
                            RT_SPHERE_PARK(T, jdx)
                            if (p_idx >= 0) resolve(p_idx, p_hb, p_disc);
                        }
                        // (the masks of the two conditions are combined as scalars, and the count is declared wave-uniform:
                        //  ballot(a && b) of two lane masks goes through a VGPR, and its popcount is compared as a vector)
                        if (RT_STEP_AT < 65 &&
                            mask_count(__builtin_amdgcn_ballot_w64(live) & ~__builtin_amdgcn_ballot_w64(k < kend)) >= RT_STEP_AT)
                            break;
                    }
                    // the cell's other primitives, one per lane and pass, for the lanes that are through with its spheres
                    if (OTHERS) {
                        while (__builtin_amdgcn_ballot_w64(live && !(k < kend) && ko < koend) != 0ull) {
                            if (live && !(k < kend) && ko < koend) {
                                const int id = (int)g_items32[ko];
                                ++ko;
                                if (COUNT) c_lane_clusters++;
                                if (id < ns + nr) {
                                    test_rect(id - ns);
                                } else {
                                    // behind the primitive's (host-grown) box: exact slab distances, as at the grid's bounds.  (The box
                                    // lies behind the primitive's records, in the same lines of memory.  Reading all of them at
                                    // once, so that the test's operands travel while the box is tested, costs 24 registers and
                                    // lost: 20000 triangles 30.5 -> 31.7 ms, the DNA frame 7.05 -> 7.47 ms.)
                                    const bool is_cyl = id < ns + nr + nc;
                                    const int kk = is_cyl ? id - ns - nr : id - ns - nr - nc;
                                    const float4 *bb = is_cyl ? cyl + RT_CYL_STRIDE * kk + 4 : tri + RT_TRI_STRIDE * kk + 3;
                                    const float4 bmn = bb[0], bmx = bb[1];
                                    const float lx = (bmn.x - ox) * bp.idx, ux = (bmx.x - ox) * bp.idx;
                                    const float ly = (bmn.y - oy) * bp.idy, uy = (bmx.y - oy) * bp.idy;
                                    const float lz = (bmn.z - oz) * bp.idz, uz = (bmx.z - oz) * bp.idz;
                                    const float tn = fmaxf(fmaxf(fmaxf(fminf(lx, ux), fminf(ly, uy)), 0.0f), fminf(lz, uz));
                                    const float tf = fminf(fminf(fminf(fmaxf(lx, ux), fmaxf(ly, uy)), best_t * 1.0001f), fmaxf(lz, uz));
                                    if (!(tn > tf)) {
                                        if (is_cyl) test_cyl(kk);
                                        else if (EXT) test_tri(kk);
                                    }
                                }
                            }
                        }
                    }
                    if (COUNT) c_groups++;
                    // The tail: a wave's walk lasts as long as its slowest lane's (11 test and 4 step passes for 2.8 tests
                    // and 0.7 steps per lane).  When only a few lanes are still walking while most of the wave waits for
                    // its shading, the stragglers stop at their next cell boundary and go on from there in the next
                    // iteration, together with the new queries (the walk is front to back: nothing nearer than the
                    // boundary was found, and whatever was found beyond it is found again in its own cell).
                    const unsigned long long walking = __builtin_amdgcn_ballot_w64(live);
                    const bool cut = RT_WALK_TAIL > 0 && mask_count(walking) <= RT_WALK_TAIL &&
                                     mask_count(__builtin_amdgcn_ballot_w64(active) & ~walking) >= RT_WALK_WAITING;
                    if (live && !(k < kend) && !(OTHERS && ko < koend)) {
                        const float tnext = SHEET ? fminf(tmx, tmz) : fminf(fminf(tmx, tmy), tmz);
                        const bool xle = tmx == tnext, yle = !SHEET && !xle && tmy == tnext;
                        const int sh = xle ? 0 : (yle ? REM_BITS : 2 * REM_BITS);
                        if (tnext > fminf(t_exit, best_t * 1.0001f) || ((rem >> sh) & REM_MASK) == 0u) {
                            live = false;
                        } else if (cut && tnext > t_from && !(best_t < tnext)) {  // (a hit inside the cell being left stays: the walk ends at the next boundary test)
                            live = false;
                            t_res = tnext;
                        } else {
                            ci += xle ? sx : (yle ? sy : sz);
                            tmx += xle ? dtx : 0.0f, tmz += (xle || yle) ? 0.0f : dtz;
                            if (!SHEET) tmy += yle ? dty : 0.0f;
                            rem -= 1u << sh;
                            cell_list(ci, k, kend);
                            if (COUNT) c_lane_cands++;
                        }
                    }
                }
                if (RT_PRIO_W != RT_PRIO_H) __builtin_amdgcn_s_setprio(RT_PRIO_H);
                // far origins that can reach the grid at all: every clustered sphere (the flat scan)
                if (__builtin_amdgcn_ballot_w64(far_scan) != 0ull) {
                    const int end = P.np + (CSIZE + 1) * P.ncl;
                    for (int i = P.np; i < end; i += 4) {
                        const float4 s0 = sph[i], s1 = sph[i + 1], s2 = sph[i + 2], s3 = sph[i + 3];
                        if (far_scan) {
                            RT_SPHERE_TEST(s0, i)
                            RT_SPHERE_TEST(s1, i + 1)
                            RT_SPHERE_TEST(s2, i + 2)
                            RT_SPHERE_TEST(s3, i + 3)
                        }
                    }
                }
                blim = best_t * 1.0001f;
            } else if (CULL == 3) {
                // windows of 64 clusters: one mask bit per cluster
                for (int w0 = 0; w0 < P.nwin; ++w0) {
                    // clip the ray to the window box (the union of its cluster boxes; same margin as every box test)
                    const BoxP bp = box_params();
                    const float idx = bp.idx, idy = bp.idy, idz = bp.idz, marg = bp.marg;
                    const float4 *wb = hot + (P.off_wbox - gap) + 2 * w0;
                    const float4 wmn = wb[0], wmx = wb[1];
                    blim = best_t * 1.0001f;  // what the prefix and the previous windows found
                    const float lx = fmaf(wmn.x, idx, bp.nxm), ux = fmaf(wmx.x, idx, bp.nxp);
                    const float ly = fmaf(wmn.y, idy, bp.nym), uy = fmaf(wmx.y, idy, bp.nyp);
                    const float lz = fmaf(wmn.z, idz, bp.nzm), uz = fmaf(wmx.z, idz, bp.nzp);
                    const float tn = fmaxf(fmaxf(fmaxf(fminf(lx, ux), fminf(ly, uy)), 0.0f), fminf(lz, uz));
                    const float tf = fminf(fminf(fminf(fmaxf(lx, ux), fmaxf(ly, uy)), blim), fmaxf(lz, uz));
                    const bool wlive = !(tn > tf);
                    if (__builtin_amdgcn_ballot_w64(wlive) == 0ull) continue;
                    if (COUNT) c_lane_groups += wlive ? 1u : 0u;
                    // phase 1: candidate clusters of the clipped segment [tn, tf]: the slab ranges of its bounding box
                    // (grown by the margin) select one precomputed mask per axis -- R_a[i0][i1] = clusters whose box
                    // overlaps the slabs i0..i1 of the window along axis a; a cluster the ray can reach overlaps the
                    // segment's box on every axis, so it is in the intersection of the three masks
                    unsigned long long cand = 0ull;
                    if (wlive) {
                        const float4 *hd = hot + (P.off_rtab - gap) + w0 * P.rt_stride;
                        const float4 gmn = hd[0], giw = hd[1];
                        const unsigned long long *tab = reinterpret_cast<const unsigned long long *>(hd + 2);
                        cand = ~0ull;
                        const float top = (float)(RT_SLABS - 1);
                        if (P.rt_axes & 1) {
                            const float a = fmaf(tn, dx, ox), b = fmaf(tf, dx, ox);
                            const int i0 = (int)__builtin_amdgcn_fmed3f(((fminf(a, b) - marg) - gmn.x) * giw.x, 0.0f, top);
                            const int i1 = (int)__builtin_amdgcn_fmed3f(((fmaxf(a, b) + marg) - gmn.x) * giw.x, 0.0f, top);
                            cand &= tab[i0 * RT_SLABS + i1];
                            tab += RT_SLABS * RT_SLABS;
                        }
                        if (P.rt_axes & 2) {
                            const float a = fmaf(tn, dy, oy), b = fmaf(tf, dy, oy);
                            const int i0 = (int)__builtin_amdgcn_fmed3f(((fminf(a, b) - marg) - gmn.y) * giw.y, 0.0f, top);
                            const int i1 = (int)__builtin_amdgcn_fmed3f(((fmaxf(a, b) + marg) - gmn.y) * giw.y, 0.0f, top);
                            cand &= tab[i0 * RT_SLABS + i1];
                            tab += RT_SLABS * RT_SLABS;
                        }
                        if (P.rt_axes & 4) {
                            const float a = fmaf(tn, dz, oz), b = fmaf(tf, dz, oz);
                            const int i0 = (int)__builtin_amdgcn_fmed3f(((fminf(a, b) - marg) - gmn.z) * giw.z, 0.0f, top);
                            const int i1 = (int)__builtin_amdgcn_fmed3f(((fmaxf(a, b) + marg) - gmn.z) * giw.z, 0.0f, top);
                            cand &= tab[i0 * RT_SLABS + i1];
                        }
                        const int left = P.ncl - w0 * 64;  // the last window may hold fewer than 64 clusters
                        if (left < 64) cand &= (1ull << left) - 1ull;
                        if (COUNT) c_lane_cands += (uint32_t)__popcll(cand);
                    }
                    // phase 2: keep the candidates whose own box the ray reaches (per-lane box reads)
                    unsigned long long mine = 0ull;
                    while (__builtin_amdgcn_ballot_w64(cand != 0ull) != 0ull) {
                        if (cand != 0ull) {
                            const unsigned long long low = cand & (0ull - cand);
                            const int q = (int)__builtin_ctzll(cand);
                            cand ^= low;
                            const float4 *b = box + 2 * (w0 * 64 + q);
                            if (slab_live(bp, b[0], b[1])) mine |= low;
                        }
                        if (COUNT) c_groups++;
                    }
                    if (COUNT) {
                        c_lane_clusters += (uint32_t)__popcll(mine);
                        uint32_t m = (uint32_t)__popcll(mine);
                        for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off, 64));
                        if (lane == 0) c_query_maxpop += m;
                    }
                    // phase 3: every lane walks its own clusters; lanes that are done wait masked off
                    while (__builtin_amdgcn_ballot_w64(mine != 0ull) != 0ull) {
                        if (mine != 0ull) {
                            const int q = (int)__builtin_ctzll(mine);
                            mine &= mine - 1ull;
                            const int base = P.np + (CSIZE + 1) * (w0 * 64 + q);
                            const float4 *cs = sph + base;
#pragma unroll
                            for (int h = 0; h < CSIZE; h += 4) {
                                const float4 r0 = cs[h], r1 = cs[h + 1], r2 = cs[h + 2], r3 = cs[h + 3];
                                RT_SPHERE_TEST(r0, base + h)
                                RT_SPHERE_TEST(r1, base + h + 1)
                                RT_SPHERE_TEST(r2, base + h + 2)
                                RT_SPHERE_TEST(r3, base + h + 3)
                            }
                        }
                        if (COUNT) c_clusters++;
                    }
                }
                blim = best_t * 1.0001f;
            } else if (CULL == 2) {
                const BoxP bp = box_params();
                // windows of 64 clusters (16 outer boxes): one mask bit per cluster
                for (int g0 = 0; g0 < P.ngr; g0 += 64 / RT_GROUP) {
                    // big scenes: one box around the whole window first (third level of the hierarchy)
                    if (P.nwin > 1) {
                        const float4 *wb = hot + (P.off_wbox - gap) + 2 * (g0 / (64 / RT_GROUP));
                        if (__builtin_amdgcn_ballot_w64(slab_live(bp, wb[0], wb[1])) == 0ull) continue;
                        blim = best_t * 1.0001f;  // what the previous windows found tightens this one
                    }
                    // phase 1: which clusters can this lane's ray reach?  (wave-uniform box reads)
                    unsigned long long mine = 0ull;
                    const int g_end = min(P.ngr, g0 + 64 / RT_GROUP);
                    for (int g = g0; g < g_end; ++g) {
                        const bool glive = slab_live(bp, gbox[2 * g], gbox[2 * g + 1]);
                        if (__builtin_amdgcn_ballot_w64(glive) == 0ull) continue;
                        if (COUNT) c_groups++, c_lane_groups += glive ? 1u : 0u;
                        const int nj = min(RT_GROUP, P.ncl - g * RT_GROUP);
                        uint32_t gm = 0;
#pragma unroll
                        for (int j = 0; j < RT_GROUP; ++j) {
                            if (j < nj) {
                                const int q = g * RT_GROUP + j;
                                if (slab_live(bp, box[2 * q], box[2 * q + 1])) gm |= 1u << j;
                            }
                        }
                        if (COUNT) c_lane_clusters += __popc(gm);
                        mine |= (unsigned long long)gm << (RT_GROUP * (g - g0));
                    }
                    if (COUNT) {
                        uint32_t m = (uint32_t)__popcll(mine);
                        for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off, 64));
                        if (lane == 0) c_query_maxpop += m;
                    }
                    // phase 2: every lane walks its own clusters; lanes that are done wait masked off
                    while (__builtin_amdgcn_ballot_w64(mine != 0ull) != 0ull) {
                        if (mine != 0ull) {
                            const int q = (int)__builtin_ctzll(mine);
                            mine &= mine - 1ull;
                            const int base = P.np + (CSIZE + 1) * (g0 * RT_GROUP + q);
                            const float4 *cs = sph + base;
                            // four records at a time: eight in flight cost 20 spilled VGPRs at 6 waves/SIMD
#pragma unroll
                            for (int h = 0; h < CSIZE; h += 4) {
                                const float4 r0 = cs[h], r1 = cs[h + 1], r2 = cs[h + 2], r3 = cs[h + 3];
                                RT_SPHERE_TEST(r0, base + h)
                                RT_SPHERE_TEST(r1, base + h + 1)
                                RT_SPHERE_TEST(r2, base + h + 2)
                                RT_SPHERE_TEST(r3, base + h + 3)
                            }
                        }
                        if (COUNT) c_clusters++;
                    }
                }
                blim = best_t * 1.0001f;
            } else if (CULL == 1) {
                const BoxP bp = box_params();
                for (int g = 0; g < P.ngr; ++g) {
                const bool glive = slab_live(bp, gbox[2 * g], gbox[2 * g + 1]);
                if (__builtin_amdgcn_ballot_w64(glive) == 0ull) continue;
                if (COUNT) c_groups++, c_lane_groups += glive ? 1u : 0u;
                const int q_end = min(P.ncl, (g + 1) * RT_GROUP);
                for (int q = g * RT_GROUP; q < q_end; ++q) {
                    const bool live = slab_live(bp, box[2 * q], box[2 * q + 1]);
                    if (COUNT && live) c_lane_clusters++;
                    if (__builtin_amdgcn_ballot_w64(live) != 0ull) {
                        const int base = P.np + (CSIZE + 1) * q;
                        const float4 *cs = sph + base;
#pragma unroll
                        for (int h = 0; h < CSIZE; h += 4) {
                            const float4 r0 = cs[h], r1 = cs[h + 1], r2 = cs[h + 2], r3 = cs[h + 3];
                            RT_SPHERE_TEST(r0, base + h)
                            RT_SPHERE_TEST(r1, base + h + 1)
                            RT_SPHERE_TEST(r2, base + h + 2)
                            RT_SPHERE_TEST(r3, base + h + 3)
                        }
                        blim = best_t * 1.0001f;
                        if (COUNT) c_clusters++;
                    }
                }
                }
            }
            }  // if (active): candidate search
            if (active) {
#undef RT_SPHERE_TEST

            // ---- rectangles, cylinders and triangles that are tested for every query: all of them in the searches without a grid
            // (the reference's loop), the oversized ones in the grid kernels -- plus, for a lane whose origin lies beyond the reach
            // of the cells' lists (far_scan), the listed ones as well
            if (!SPH) {
                const bool any_far = GRID && __builtin_amdgcn_ballot_w64(far_scan) != 0ull;
                const int nr_loop = (GRID && !any_far) ? P.nr_a : nr, nc_loop = (GRID && !any_far) ? P.nc_a : nc;
                const int nt_loop = EXT ? ((GRID && !any_far) ? P.nt_a : nt) : 0;
                for (int j = 0; j < nr_loop; ++j)
                    if (!GRID || j < P.nr_a || far_scan) test_rect(j);
                // the boxes of cylinders and triangles: the box-test values once more (see box_params)
                BoxP bq = {};
                if (CULL && (nc_loop > 0 || nt_loop > 0)) bq = box_params();
                for (int k = 0; k < nc_loop; ++k) {
                    const bool mine = !GRID || k < P.nc_a || far_scan;
                    if (CULL) {
                        blim = best_t * 1.0001f;  // the cylinder's world-space box, same margin (the object-space quadratic has the
                                 // same error structure as the sphere test: ~1e-3 |o| in space)
                        const float4 *cb = cyl + RT_CYL_STRIDE * k + 4;
                        if (__builtin_amdgcn_ballot_w64(mine && slab_live(bq, cb[0], cb[1])) == 0ull) continue;
                    }
                    if (mine) test_cyl(k);
                }
                for (int k = 0; k < nt_loop; ++k) {
                    const bool mine = !GRID || k < P.nt_a || far_scan;
                    if (CULL) {
                        blim = best_t * 1.0001f;
                        const float4 *tb = tri + RT_TRI_STRIDE * k + 3;
                        if (__builtin_amdgcn_ballot_w64(mine && slab_live(bq, tb[0], tb[1])) == 0ull) continue;
                    }
                    if (mine) test_tri(k);
                }
            }
            // (a lane whose grid walk was cut short has no result yet: its query goes on in the next iteration)
            const bool unfinished = GRID && t_res != 0.0f;
            if (COUNT) {
                if (!unfinished) c_queries++;
                const unsigned long long alive = __builtin_amdgcn_ballot_w64(true);
                if ((int)__builtin_ctzll(alive) == lane) {
                    c_wave_queries++;
                    atomicAdd(&counters->occ_hist[queue_empty ? 1 : 0][__popcll(alive) >> 2], 1ull);
                }
            }

            tick(2);
            // ---- (2) the winner (ray_color body, main.cu:45-65 / main.cpp:22-38)
            // 1/|d| once per query (metal, dielectric and the sky all normalise the direction)
            inv_len = 1.0f / rt_sqrtf(ra);
            if (unfinished) {
            } else if (best_id >= 0) {
                // hit record of the winner only (the reference fills one per candidate)
                if (SPH || best_id < ns) {
                    const float4 s = sph[best_id];
                    const float4 cold = image[P.off_sph_cold + best_id];
                    px = fmaf(best_t, dx, ox), py = fmaf(best_t, dy, oy), pz = fmaf(best_t, dz, oz);
                    const float onx = cold.x * (px - s.x), ony = cold.x * (py - s.y), onz = cold.x * (pz - s.z);
                    front = dot3(dx, dy, dz, onx, ony, onz) < 0.0f;
                    nx = front ? onx : -onx, ny = front ? ony : -ony, nz = front ? onz : -onz;
                    mat = __float_as_int(cold.y);
                    kind = __float_as_int(cold.w);
                } else if (best_id < ns + nr) {
                    const int j = best_id - ns;
                    const int axis = __float_as_int(rect[2 * j + 1].y);
                    px = fmaf(best_t, dx, ox), py = fmaf(best_t, dy, oy), pz = fmaf(best_t, dz, oz);
                    const float dk = axis == 0 ? dz : (axis == 1 ? dy : dx);
                    front = dk < 0.0f;
                    // front ? (0,0,1) : -(0,0,1), zeros keep their sign as in the reference
                    const float sgn = front ? 1.0f : -1.0f, zer = front ? 0.0f : -0.0f;
                    nx = axis == 2 ? sgn : zer, ny = axis == 1 ? sgn : zer, nz = axis == 0 ? sgn : zer;
                    const float4 rc = image[P.off_rect_cold + j];
                    mat = __float_as_int(rc.x);
                    kind = __float_as_int(rc.z);
                } else if (best_id < ns + nr + nc) {
                    const int k = best_id - ns - nr;
                    const float4 r0 = cyl[RT_CYL_STRIDE * k], r1 = cyl[RT_CYL_STRIDE * k + 1], r2 = cyl[RT_CYL_STRIDE * k + 2];
                    const float4 *cc4 = image + P.off_cyl_cold + 4 * k;
                    const float4 m0 = cc4[0], m1 = cc4[1], m2 = cc4[2];
                    const float oox = fmaf(r0.x, ox, fmaf(r0.y, oy, fmaf(r0.z, oz, r0.w)));
                    const float ooy = fmaf(r1.x, ox, fmaf(r1.y, oy, fmaf(r1.z, oz, r1.w)));
                    const float ooz = fmaf(r2.x, ox, fmaf(r2.y, oy, fmaf(r2.z, oz, r2.w)));
                    const float odx = fmaf(r0.x, dx, fmaf(r0.y, dy, r0.z * dz));
                    const float ody = fmaf(r1.x, dx, fmaf(r1.y, dy, r1.z * dz));
                    const float odz = fmaf(r2.x, dx, fmaf(r2.y, dy, r2.z * dz));
                    const float opx = fmaf(best_t, odx, oox), opy = fmaf(best_t, ody, ooy), opz = fmaf(best_t, odz, ooz);
                    const float len = rt_sqrtf(fmaf(opx, opx, opy * opy));
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
                    kind = __float_as_int(cc4[3].z);
                } else if (EXT) {  // triangle, taichi-version/hittable.py:254-259: the stored unit normal, turned against the ray
                    const int k = best_id - ns - nr - nc;
                    const float tnx = tri[RT_TRI_STRIDE * k].w, tny = tri[RT_TRI_STRIDE * k + 1].w, tnz = tri[RT_TRI_STRIDE * k + 2].w;
                    px = fmaf(best_t, dx, ox), py = fmaf(best_t, dy, oy), pz = fmaf(best_t, dz, oz);
                    front = dot3(dx, dy, dz, tnx, tny, tnz) < 0.0f;
                    nx = front ? tnx : -tnx, ny = front ? tny : -tny, nz = front ? tnz : -tnz;
                    mat = __float_as_int(image[P.off_tri_cold + 2 * k].x);
                    kind = __float_as_int(image[P.off_mat + 3 * mat].x);
                }
                // (the material kind rides in the primitive's cold record: one dependent load, not two)
                const float4 *M = image + P.off_mat + 3 * mat;
                // the hit record's (u, v) -- only where the material's texture reads them (an image texture); every
                // other texture of the reference ignores them, and acos / atan2 per candidate hit (object.cuh:87-93)
                // would be the most expensive part of sphere::hit
                if (EXT && (kind == MK_LAMBERT_IMAGE || kind == MK_LIGHT_IMAGE)) {
                    float tu, tv;
                    if (best_id < ns) {  // get_sphere_uv(outward_normal), object.cuh:87-93
                        const float onx = front ? nx : -nx, ony = front ? ny : -ny, onz = front ? nz : -nz;
                        const float theta = rt_acosf(-ony);
                        const float phi = rt_atan2f(-onz, onx) + 3.1415927410125732421875f;
                        tu = phi / 6.283185482025146484375f;
                        tv = theta / 3.1415927410125732421875f;
                    } else if (best_id < ns + nr) {  // object.cuh:113-114, 150-151, 183-184
                        const int j = best_id - ns;
                        const float4 q0r = rect[2 * j];
                        const int axis = __float_as_int(rect[2 * j + 1].y);
                        const float pa = axis == 2 ? py : px, pb = axis == 0 ? py : pz;
                        tu = (pa - q0r.x) / (q0r.y - q0r.x);
                        tv = (pb - q0r.z) / (q0r.w - q0r.z);
                    } else if (best_id < ns + nr + nc) {  // object.cuh:283-288, in the cylinder's object space
                        const int k = best_id - ns - nr;
                        const float4 r0 = cyl[RT_CYL_STRIDE * k], r1 = cyl[RT_CYL_STRIDE * k + 1], r2 = cyl[RT_CYL_STRIDE * k + 2], pr = cyl[RT_CYL_STRIDE * k + 3];
                        const float oox = fmaf(r0.x, ox, fmaf(r0.y, oy, fmaf(r0.z, oz, r0.w)));
                        const float ooy = fmaf(r1.x, ox, fmaf(r1.y, oy, fmaf(r1.z, oz, r1.w)));
                        const float ooz = fmaf(r2.x, ox, fmaf(r2.y, oy, fmaf(r2.z, oz, r2.w)));
                        const float odx = fmaf(r0.x, dx, fmaf(r0.y, dy, r0.z * dz));
                        const float ody = fmaf(r1.x, dx, fmaf(r1.y, dy, r1.z * dz));
                        const float odz = fmaf(r2.x, dx, fmaf(r2.y, dy, r2.z * dz));
                        const float opx = fmaf(best_t, odx, oox), opy = fmaf(best_t, ody, ooy), opz = fmaf(best_t, odz, ooz);
                        const float phi = rt_atan2f(opy, opx) + 6.283185482025146484375f;
                        tu = phi / 12.56637096405029296875f;
                        tv = (opz - pr.y) / (pr.z - pr.y);
                    } else {  // hittable.py:54-58, 233: area weights of the plane point, uv = u1 w1 + u2 w2 + u3 w3
                        const int k = best_id - ns - nr - nc;
                        const float4 r0 = tri[RT_TRI_STRIDE * k], r1 = tri[RT_TRI_STRIDE * k + 1], r2 = tri[RT_TRI_STRIDE * k + 2];
                        float rix, riy, riz, root;
                        tri_plane(r0, r1, r2, rix, riy, riz, root);
                        const float a1x = rix - r0.x, a1y = riy - r0.y, a1z = riz - r0.z;
                        const float a2x = rix - r1.x, a2y = riy - r1.y, a2z = riz - r1.z;
                        const float a3x = rix - r2.x, a3y = riy - r2.y, a3z = riz - r2.z;
                        float cx, cy, cz, ex, ey, ez;
                        cross3(a1x, a1y, a1z, a2x, a2y, a2z, cx, cy, cz);
                        cross3(r2.x - r0.x, r2.y - r0.y, r2.z - r0.z, r2.x - r1.x, r2.y - r1.y, r2.z - r1.z, ex, ey, ez);
                        const float w1 = rt_sqrtf(dot3(cx, cy, cz, cx, cy, cz)) / rt_sqrtf(dot3(ex, ey, ez, ex, ey, ez));
                        cross3(a1x, a1y, a1z, a3x, a3y, a3z, cx, cy, cz);
                        cross3(r1.x - r0.x, r1.y - r0.y, r1.z - r0.z, r1.x - r2.x, r1.y - r2.y, r1.z - r2.z, ex, ey, ez);
                        const float w2 = rt_sqrtf(dot3(cx, cy, cz, cx, cy, cz)) / rt_sqrtf(dot3(ex, ey, ez, ex, ey, ez));
                        cross3(a3x, a3y, a3z, a2x, a2y, a2z, cx, cy, cz);
                        cross3(r0.x - r2.x, r0.y - r2.y, r0.z - r2.z, r0.x - r1.x, r0.y - r1.y, r0.z - r1.z, ex, ey, ez);
                        const float w3 = rt_sqrtf(dot3(cx, cy, cz, cx, cy, cz)) / rt_sqrtf(dot3(ex, ey, ez, ex, ey, ez));
                        const float4 c0 = image[P.off_tri_cold + 2 * k], c1 = image[P.off_tri_cold + 2 * k + 1];
                        tu = fmaf(c1.z, w3, fmaf(c1.x, w2, c0.z * w1));
                        tv = fmaf(c1.w, w3, fmaf(c1.y, w2, c0.w * w1));
                    }
                    image_texel(image, M[1], tu, tv, tex_r, tex_g, tex_b);
                }
                if (COUNT) {
                    c_hits++;
                    if (kind <= MK_LAMBERT_IMAGE) c_scatter0++;
                    else if (kind == MK_METAL) c_scatter1++;
                    else if (kind == MK_DIELECTRIC) c_scatter2++;
                    else c_scatter3++;
                }
                if (kind >= MK_LIGHT_SOLID) {  // diffuse_light: emitted, never scatters (material.cuh:161-182, main.cu:48-58)
                    const float4 q1 = M[1], q2 = M[2];
                    const bool odd = kind == MK_LIGHT_CHECKER && checker_odd(px, py, pz);
                    float er = odd ? q2.x : q1.x, eg = odd ? q2.y : q1.y, eb = odd ? q2.z : q1.z;
                    if (EXT && kind == MK_LIGHT_IMAGE) er = tex_r, eg = tex_g, eb = tex_b;
                    L_r = er * beta_r, L_g = eg * beta_g, L_b = eb * beta_b;
                    path_done = true;  // absorbed: main.cu:55-58
                    kind = -1;
                }
            } else {
                // miss: main.cpp:36-38 (sky) or main.cu:63 (constant background)
                float bg_r, bg_g, bg_b;
                if (P.flags & RT_FLAG_SKY_GRADIENT) {
                    const float t = 0.5f * (inv_len * dy + 1.0f);
                    const float omt = 1.0f - t;
                    bg_r = fmaf(t, 0.5f, omt), bg_g = fmaf(t, 0.7f, omt), bg_b = fmaf(t, 1.0f, omt);
                } else {
                    bg_r = P.background[0], bg_g = P.background[1], bg_b = P.background[2];
                }
                L_r = beta_r * bg_r, L_g = beta_g * bg_g, L_b = beta_b * bg_b;
                path_done = true;
                if (COUNT) c_misses++;
            }
            }  // if (active): rects, cylinders, triangles, shading part 1
          }
        }
        tick(3);
        // ---- (3) res += ray_color(...), main.cu:100 -- exact fixed-point add into the tile
        if (path_done) {
            {
                const unsigned long long fr = radiance_to_fixed(L_r), fg = radiance_to_fixed(L_g), fb = radiance_to_fixed(L_b);
                if (cur_p < 0) {
                    unsigned long long *g = acc + (size_t)(~cur_p) * 3;
                    if (fr) atomicAdd(g + 0, fr);
                    if (fg) atomicAdd(g + 1, fg);
                    if (fb) atomicAdd(g + 2, fb);
                } else {
                    unsigned long long *a = c_acc + cur_p * 3;
                    atomicAdd(a + 0, fr);
                    atomicAdd(a + 1, fg);
                    atomicAdd(a + 2, fb);
                }
            }
            active = false;
        }
        tick(4);
        if (RT_PRIO_H != RT_PRIO_F) __builtin_amdgcn_s_setprio(RT_PRIO_F);
        // ---- (4) refill: lanes without a live path take new samples
        // (render()'s sample loop, main.cu:95-101; camera::get_ray camera.h:32-39)
        float u = 0, v = 0;    // jitter of the sample a lane starts (main.cu:96-97)
        bool started = false;  // this lane starts a new path in this iteration
        const bool need = !active;
        const unsigned long long idle = __ballot(need);
        // is the current item handed out completely?
        bool exhausted = !c_valid;
        if (c_valid) {
            if (POOL) exhausted = cursor >= c_pool;
            else exhausted = __builtin_amdgcn_ballot_w64(c_hvalid != 0 && mine * 64 < c_pool) == 0ull;
        }
        // The pool is handed out and idle lanes want the next item: retire the current one.  Its
        // accumulator is flushed for reuse and every path still alive becomes an orphan (three 64-bit global atomics
        // when it ends: one more dirty 64-byte line for 24 useful bytes).  So the item is only retired once at most
        // P.orphan_max paths are left, the idle lanes wait meanwhile.  Measured 12 / 32 / 63 (never wait): whole frame
        // 147.9 / 147.9 / 147.3 ms and 1.16 / - / 2.0 GB of HBM writes, a 1/8 row shard (short items) 20.36 / 20.11 /
        // 19.97 ms: the host sets 12 for launches with many tiles per wave and 63 for small ones.
        bool fetch = exhausted && !queue_empty;
        if (c_valid && idle != 0ull && fetch) {
            if (mask_count(~idle) <= P.orphan_max) {
                if (active && cur_p >= 0) cur_p = ~((c_band * 8 + (cur_p >> 3)) * P.width + c_x0 + (cur_p & 7));
                flush_tile(c_acc, c_x0, c_band);
                c_valid = false;
            } else {
                fetch = false;
            }
        }
        if (idle) {  // wave-uniform
            if (fetch) {
                unsigned int item = 0;
                if (lane == 0) item = atomicAdd(queue, 1u);
                item = __builtin_amdgcn_readfirstlane(item);
                const int4 ia = ipar4(0), ib = ipar4(1), ic = ipar4(2);
                if (item >= (unsigned int)ia.z) {
                    queue_empty = true;  // the counter only grows: every wave gets here
                    if (COUNT) t_qe = __builtin_amdgcn_s_memrealtime();
                } else {
                    const unsigned int tiles_x = (unsigned int)ia.x, bands = (unsigned int)ia.y;
                    const int sample_first = ia.w, sample_count = ib.x, spp_chunk = ib.y;
                    const int n_big = ib.z, n_med = ib.w, q_med = ic.x, q_small = ic.y;
                    c_x0 = (int)(item % tiles_x) * 8;
                    c_band = (int)((item / tiles_x) % bands);
                    const int chunk = (int)(item / (tiles_x * bands));
                    int s_stop;  // sample range: big chunks first, shorter and shorter ones towards the end of the queue
                    if (chunk < n_big) {
                        c_sbegin = sample_first + chunk * spp_chunk;
                        s_stop = c_sbegin + spp_chunk;
                    } else if (chunk < n_big + n_med) {
                        c_sbegin = sample_first + n_big * spp_chunk + (chunk - n_big) * q_med;
                        s_stop = c_sbegin + q_med;
                    } else {
                        c_sbegin = sample_first + n_big * spp_chunk + n_med * q_med + (chunk - n_big - n_med) * q_small;
                        s_stop = c_sbegin + q_small;
                    }
                    const int s_end = sample_first + sample_count;
                    if (s_stop > s_end) s_stop = s_end;
                    c_pool = (s_stop - c_sbegin) * 64;  // pool item k = (pixel k & 63, sample c_sbegin + (k >> 6))
                    // (the item's values are the same in every lane; said so, they live in scalar registers instead of five of
                    //  the 72 vector registers this kernel spills from)
                    c_x0 = __builtin_amdgcn_readfirstlane(c_x0), c_band = __builtin_amdgcn_readfirstlane(c_band);
                    c_sbegin = __builtin_amdgcn_readfirstlane(c_sbegin), c_pool = __builtin_amdgcn_readfirstlane(c_pool);
                    cursor = 0;
                    mine = 0;
                    c_valid = true;
                    int hx_unused, hlr_unused;
                    home_pixel(c_x0, c_band, hx_unused, hlr_unused, c_hy, c_hvalid);
                    c_hvmask = __builtin_amdgcn_ballot_w64(c_hvalid != 0);
                }
            }
            bool start = false;
            int sp = 0, spx = 0, spy = 0, ss = 0;
            if (POOL) {
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32),
                                                                __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                const int k = cursor + rank;
                if (c_valid) cursor = min(cursor + mask_count(idle), c_pool);
                sp = k & 63;
                // row and validity of pixel sp live in lane sp's registers (all lanes take part)
                spy = __shfl(c_hy, sp, 64);
                // (on-image bit of pixel sp from the item's lane mask: a select and a bit-field extract instead of a second
                //  cross-lane read)
                const uint32_t hv_word = (sp & 32) ? (uint32_t)(c_hvmask >> 32) : (uint32_t)c_hvmask;
                const int pv = (int)((hv_word >> (sp & 31)) & 1u);
                spx = c_x0 + (sp & 7);
                ss = c_sbegin + (k >> 6);
                start = need && c_valid && k < c_pool && pv != 0;
            } else {
                start = need && c_valid && mine * 64 < c_pool && c_hvalid != 0;
                sp = lane, spx = c_x0 + (lane & 7), spy = c_hy, ss = c_sbegin + mine;
                if (start) mine++;
            }
            if (start) {
                cur_p = sp;
                rng_start(rng, (uint32_t)(spy * P.width + spx), (uint32_t)ss, k0, k1);
                u = ((float)spx + rng_next<COUNT>(rng)) * P.inv_wm1;
                v = ((float)spy + rng_next<COUNT>(rng)) * P.inv_hm1;
                if (COUNT) c_samples++;
            }
            started = start;
        }
        tick(0);
        if (!__any(active || started)) {
            // nothing in flight.  Out of work when the queue is dry, the current item is handed out and no sample
            // waits to be added; otherwise loop: the refill above makes progress every time (takes an item,
            // marks the queue empty, or skips off-image pool entries).
            bool exhausted = !c_valid;
            if (c_valid) {
                if (POOL) exhausted = cursor >= c_pool;
                else exhausted = __builtin_amdgcn_ballot_w64(c_hvalid != 0 && mine * 64 < c_pool) == 0ull;
            }
            if (queue_empty && exhausted) {
                if (c_valid) flush_tile(c_acc, c_x0, c_band);
                break;
            }
            continue;
        }
        // ---- (5) rejection sampling, one converged loop: random_in_unit_sphere (vec3.h:121-129: three draws, for the
        // lanes whose material scatters with one: lambertian, metal) and random_in_unit_disk (vec3.h:157-165: two
        // draws, for the lens sample of the paths that start)
        const bool need_s = kind >= 0 && kind <= MK_METAL;
        const bool need_d = started && (P.flags & RT_FLAG_DEFOCUS_BLUR) != 0u;
        float sx = 0, sy = 0, sz = 0, sl2 = 1;
        if (need_s || need_d) {
            if (RT_PRIO_R != RT_PRIO_F) __builtin_amdgcn_s_setprio(RT_PRIO_R);
            // The three draws of an attempt written out on the generator's four state words (xor128_next, philox.h): every
            // lane computes the third value, and the lanes that sample a disk do not keep it -- their state advances by two
            // draws, the others' by three, through four selects on a loop-invariant mask.  With the third draw behind a
            // branch the loop carried the rotating state through ten register moves per pass: 46 VALU per pass, 39 now.
            // (Measured twice: before the wave priorities it was 0.5 % SLOWER than the branch, with them 1.7 % faster.)
            uint32_t x = rng.g.x, y = rng.g.y, z = rng.g.z, w = rng.g.w;
            do {
                const uint32_t tx = x ^ (x << 11), ty = y ^ (y << 11), tz = z ^ (z << 11);
                const uint32_t n1 = (w ^ (w >> 19)) ^ (tx ^ (tx >> 8));
                const uint32_t n2 = (n1 ^ (n1 >> 19)) ^ (ty ^ (ty >> 8));
                const uint32_t n3 = (n2 ^ (n2 >> 19)) ^ (tz ^ (tz >> 8));
                sx = fmaf((float)(n1 >> 8), 1.0f / 8388608.0f, -1.0f);
                sy = fmaf((float)(n2 >> 8), 1.0f / 8388608.0f, -1.0f);
                const float s3 = fmaf((float)(n3 >> 8), 1.0f / 8388608.0f, -1.0f);
                sz = need_s ? s3 : 0.0f;
                x = need_s ? w : z, y = need_s ? n1 : w, z = need_s ? n2 : n1, w = need_s ? n3 : n2;
                if (COUNT) rng.draws += need_s ? 3u : 2u;
                sl2 = dot3(sx, sy, sz, sx, sy, sz);  // disk: fma(x, x, y * y) -- the product with sz = 0 adds an exact zero
            } while (sl2 >= 1.0f);
            rng.g.x = x, rng.g.y = y, rng.g.z = z, rng.g.w = w;
            if (RT_PRIO_R != RT_PRIO_F) __builtin_amdgcn_s_setprio(RT_PRIO_F);
        }
        if (RT_PRIO_S != RT_PRIO_F) __builtin_amdgcn_s_setprio(RT_PRIO_S);
        // ---- (6a) the scatter step of the paths that go on
        bool fresh = false;  // this lane has a new ray
        if (kind >= 0) {
            const float4 *M = image + P.off_mat + 3 * mat;
            const float4 q0 = M[0], q1 = M[1], q2 = M[2];
                float ndx, ndy, ndz;           // scattered direction
                float at_r, at_g, at_b;        // attenuation
                bool scattered = true;
                if (kind <= MK_LAMBERT_IMAGE) {  // lambertian::scatter, material.h:25-35
                    const float inv = 1.0f / rt_sqrtf(sl2);
                    ndx = nx + inv * sx, ndy = ny + inv * sy, ndz = nz + inv * sz;
                    const float eps = 1e-8f;
                    if (fabsf(ndx) < eps && fabsf(ndy) < eps && fabsf(ndz) < eps) ndx = nx, ndy = ny, ndz = nz;
                    const bool odd = kind == MK_LAMBERT_CHECKER && checker_odd(px, py, pz);
                    at_r = odd ? q2.x : q1.x, at_g = odd ? q2.y : q1.y, at_b = odd ? q2.z : q1.z;
                    if (EXT && kind == MK_LAMBERT_IMAGE) at_r = tex_r, at_g = tex_g, at_b = tex_b;
                } else if (kind == MK_METAL) {  // metal::scatter, material.h:47-53
                    const float ux = inv_len * dx, uy = inv_len * dy, uz = inv_len * dz;
                    const float k2 = 2.0f * dot3(ux, uy, uz, nx, ny, nz);
                    const float rx = fmaf(-k2, nx, ux), ry = fmaf(-k2, ny, uy), rz = fmaf(-k2, nz, uz);
                    ndx = fmaf(q0.y, sx, rx), ndy = fmaf(q0.y, sy, ry), ndz = fmaf(q0.y, sz, rz);
                    at_r = q1.x, at_g = q1.y, at_b = q1.z;
                    scattered = dot3(ndx, ndy, ndz, nx, ny, nz) > 0.0f;
                } else {  // dielectric::scatter, material.h:66-95
                    const float ratio = front ? q0.z : q0.y;
                    const float ux = inv_len * dx, uy = inv_len * dy, uz = inv_len * dz;
                    const float udn = dot3(ux, uy, uz, nx, ny, nz);
                    const float cos_t = fminf(-udn, 1.0f);
                    const float sin_t = rt_sqrtf(fmaf(-cos_t, cos_t, 1.0f));
                    bool refl = ratio * sin_t > 1.0f;
                    if (!refl) {
                        const float r0 = front ? q0.w : q1.w;
                        const float xx = 1.0f - cos_t;
                        const float x2 = xx * xx;
                        const float x5 = (x2 * x2) * xx;
                        refl = fmaf(1.0f - r0, x5, r0) > rng_next<COUNT>(rng);
                    }
                    if (refl) {  // reflect(), vec3.h:144-147
                        const float k2 = 2.0f * udn;
                        ndx = fmaf(-k2, nx, ux), ndy = fmaf(-k2, ny, uy), ndz = fmaf(-k2, nz, uz);
                    } else {  // refract(), vec3.h:149-155
                        const float ppx = ratio * fmaf(cos_t, nx, ux);
                        const float ppy = ratio * fmaf(cos_t, ny, uy);
                        const float ppz = ratio * fmaf(cos_t, nz, uz);
                        const float kk = -rt_sqrtf(fabsf(1.0f - dot3(ppx, ppy, ppz, ppx, ppy, ppz)));
                        ndx = fmaf(kk, nx, ppx), ndy = fmaf(kk, ny, ppy), ndz = fmaf(kk, nz, ppz);
                    }
                    at_r = at_g = at_b = 1.0f;
                }

            if (scattered) {
                beta_r *= at_r, beta_g *= at_g, beta_b *= at_b;
                ox = px, oy = py, oz = pz;
                dx = ndx, dy = ndy, dz = ndz;
                depth--;
                fresh = true;
                if (depth <= 0) active = false;  // main.cpp:42 / main.cu:69: black
            } else {
                active = false;  // absorbed: main.cpp:32 / main.cu:55-58: black
            }
        }
        // ---- (6b) the camera ray of the paths that start (camera::get_ray, camera.h:32-39)
        if (started) {
                float offx = 0.0f, offy = 0.0f, offz = 0.0f;
                // the camera's derived vectors come from the hot table (wave-uniform reads, used here only),
                // not from kernel arguments that would sit in SGPRs for the whole launch
                const float4 *cv = hot + P.off_cam;
                const float4 c_org = cv[0];  // origin, lens_radius
                if (P.flags & RT_FLAG_DEFOCUS_BLUR) {
                    const float4 c_u = cv[4], c_v = cv[5];
                    float rdx = c_org.w * sx, rdy = c_org.w * sy;
                    offx = fmaf(c_u.x, rdx, c_v.x * rdy);
                    offy = fmaf(c_u.y, rdx, c_v.y * rdy);
                    offz = fmaf(c_u.z, rdx, c_v.z * rdy);
                }
                const float4 c_ll = cv[1], c_hor = cv[2], c_ver = cv[3];
                dx = fmaf(v, c_ver.x, fmaf(u, c_hor.x, c_ll.x));
                dy = fmaf(v, c_ver.y, fmaf(u, c_hor.y, c_ll.y));
                dz = fmaf(v, c_ver.z, fmaf(u, c_hor.z, c_ll.z));
                dx = (dx - c_org.x) - offx;
                dy = (dy - c_org.y) - offy;
                dz = (dz - c_org.z) - offz;
                ox = c_org.x + offx;
                oy = c_org.y + offy;
                oz = c_org.z + offz;
                beta_r = beta_g = beta_b = 1.0f;
                depth = P.max_depth;
                active = true;
                fresh = true;
        }
        // ---- (6c) what every new ray needs, however it came about
        if (fresh) {
            ra = dot3(dx, dy, dz, dx, dy, dz);
            rinv_a = 1.0f / ra;
        }
        // Russian roulette before the next query (4_0_path_tracing.py:45-46; include/rtmi.h,
        // rt_scene_set_russian_roulette): a path that does not survive keeps what it has collected (a new one:
        // nothing, so there is nothing to add); a survivor's throughput is divided by p at once
        if (P.rr_p > 0.0f) {
            if (started) {
                if (rng_next<COUNT>(rng) > P.rr_p) active = false;
                beta_r = beta_g = beta_b = 1.0f / P.rr_p;
            } else if (fresh && active) {
                if (rng_next<COUNT>(rng) > P.rr_p) active = false;
                beta_r = beta_r / P.rr_p, beta_g = beta_g / P.rr_p, beta_b = beta_b / P.rr_p;
            }
        }
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
        wave_add(&counters->cand_lanes, c_cand);
        wave_add(&counters->cand_waves, c_cand_wave);
        if (lane == 0 && c_clusters) atomicAdd(&counters->clusters_visited, (unsigned long long)c_clusters);
        if (lane == 0 && c_groups) atomicAdd(&counters->groups_visited, (unsigned long long)c_groups);
        wave_add(&counters->lane_clusters, c_lane_clusters);
        wave_add(&counters->lane_groups, c_lane_groups);
        wave_add(&counters->lane_cands, c_lane_cands);
        wave_add(&counters->group_maxpop, c_group_maxpop);
        wave_add(&counters->query_maxpop, c_query_maxpop);
        if (lane == 0) {
            for (int i = 0; i < 6; ++i) atomicAdd(&counters->cycles[i], cyc[i]);
            const unsigned long long t = __builtin_amdgcn_s_memrealtime();
            atomicMin(&counters->t_end_min, t);
            atomicMax(&counters->t_end_max, t);
            atomicAdd(&counters->life_cycles, __builtin_amdgcn_s_memtime() - c_begin);
            atomicAdd(&counters->life_ticks, t - t_begin);
            atomicMin(&counters->t_qe_min, t_qe);
            atomicMax(&counters->t_qe_max, t_qe);
            const unsigned long long bin = (t - t_qe) / 5000ull;  // 100 MHz ticks -> 50 us bins
            atomicAdd(&counters->drain_hist[bin < 31 ? bin : 31], 1u);
            const unsigned long long b1 = (t_qe - t_begin) / 6400ull, b2 = (t - t_begin) / 6400ull;
            atomicAdd(&counters->qe_hist[b1 < 1023 ? b1 : 1023], 1u);
            atomicAdd(&counters->exit_hist[b2 < 1023 ? b2 : 1023], 1u);
        }
        wave_add(&counters->wave_queries, c_wave_queries);
    }
}

// fixed-point pixel sums -> fp32 framebuffer (rgb_sum[(row*W + x)*3 + c]); every store
// instruction writes 256 contiguous bytes
__global__ __launch_bounds__(256) void finalize_kernel(const unsigned long long *__restrict__ acc,
                                                       float *__restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (float)((double)(long long)acc[i] * (1.0 / 16777216.0));
}

#ifdef RT_ISA_ONLY
// tools/isa_stats.py: one instance alone (RT_ISA_ONLY = its template arguments), compiled to assembly in seconds
template __global__ void render_kernel<RT_ISA_ONLY>(const RenderParams, const float4 *__restrict__, unsigned long long *__restrict__,
                                                    unsigned int *__restrict__, DevCounters *__restrict__);
#else
// ---------------------------------------------------------------- launchers used by render_host.hip
// X(variant id, POOL, SCALAR, CULL, SPH).  The host resolves variant 0 to one of the five PRODUCT instances:
//    2  compact grid one cell high, tables in LDS (RTIOW)         6  compact grid, 3-D walk, tables in LDS (sphere-only scenes)
//   36  wide grid tables in LDS (scenes with other primitives)   44  wide grid tables in global memory (large scenes)
//   16  no culling -- the reference's linear hittable_list scan: what a scene of a handful of primitives of several types runs
//       (nothing is listed in a grid there; sample_scene.json 54 ms against 79 ms for 36 with its empty grid), and the
//       definition every other kernel's image is held to
// and 36, 44 and 16 also exist with EXT (triangles, image textures): eight render_kernel instances in a product build
// (make ABLATIONS=0).  The default build (RTMI_ABLATIONS=1: tests, bench.py) adds the measurement variants -- same image, bit for
// bit -- and the counting kernels:
//    1  variant 6 with strict one-lane-per-pixel ownership       40  variant 6 with its tables in global memory
//   17  variant 16 with strict ownership                         24  variant 16 with its tables in global memory
//   32  wave-level cluster votes      64  per-lane cluster lists through the two-level box hierarchy (round 1's default)
//  128  per-lane cluster lists through the range tables (first half of round 2)
#define RT_PRODUCT_TABLE(X)          \
    X(2, true, false, 6, true)       \
    X(6, true, false, 5, true)       \
    X(36, true, false, 7, false)     \
    X(44, true, true, 7, false)      \
    X(16, true, false, 0, false)
#define RT_PRODUCT_EXT_TABLE(X)      \
    X(36, true, false, 7, false)     \
    X(44, true, true, 7, false)      \
    X(16, true, false, 0, false)
#if RTMI_ABLATIONS
#define RT_ABLATION_TABLE(X)         \
    X(1, false, false, 5, true)      \
    X(40, true, true, 5, true)       \
    X(17, false, false, 0, false)    \
    X(24, true, true, 0, false)      \
    X(32, true, false, 1, false)     \
    X(64, true, false, 2, false)     \
    X(128, true, false, 3, false)
#define RT_ABLATION_EXT_TABLE(X) X(24, true, true, 0, false)
// counting kernels (rt_render_hip_count): X(variant, SCALAR, CULL, EXT, SPH)
#define RT_COUNT_TABLE(X)            \
    X(6, false, 5, false, true)      \
    X(36, false, 7, true, false)     \
    X(44, true, 7, true, false)      \
    X(64, false, 2, false, false)    \
    X(128, false, 3, false, false)
#else
#define RT_ABLATION_TABLE(X)
#define RT_ABLATION_EXT_TABLE(X)
#define RT_COUNT_TABLE(X)
#endif
#define RT_VARIANT_TABLE(X) RT_PRODUCT_TABLE(X) RT_ABLATION_TABLE(X)
#define RT_EXT_TABLE(X) RT_PRODUCT_EXT_TABLE(X) RT_ABLATION_EXT_TABLE(X)

bool has_ablations() { return RTMI_ABLATIONS != 0; }

// launches the instance of a RESOLVED variant (never 0); false: no such build
bool launch_render(const RenderParams &P, const void *image, unsigned long long *acc, unsigned int *queue,
                   DevCounters *counters, size_t lds_bytes, unsigned grid, hipStream_t stream, unsigned variant, bool ext) {
    const float4 *img = (const float4 *)image;
    const dim3 g(grid), t(256);
    if (counters) {
#define RT_LAUNCH_COUNT(V, SCALAR, CULL, EXT, SPH)                                                                                         \
    if (variant == V) {                                                                                                                     \
        hipLaunchKernelGGL((render_kernel<true, true, SCALAR, CULL, EXT, SPH>), g, t, lds_bytes, stream, P, img, acc, queue, counters);   \
        return true;                                                                                                                        \
    }
        RT_COUNT_TABLE(RT_LAUNCH_COUNT)
#undef RT_LAUNCH_COUNT
        return false;
    }
    DevCounters *none = nullptr;
    if (ext) {
#define RT_LAUNCH_EXT(V, POOL, SCALAR, CULL, SPH)                                                                                          \
    if (variant == V) {                                                                                                                     \
        hipLaunchKernelGGL((render_kernel<false, POOL, SCALAR, CULL, true, false>), g, t, lds_bytes, stream, P, img, acc, queue, none);   \
        return true;                                                                                                                        \
    }
        RT_EXT_TABLE(RT_LAUNCH_EXT)
#undef RT_LAUNCH_EXT
        return false;
    }
#define RT_LAUNCH(V, POOL, SCALAR, CULL, SPH)                                                                                              \
    if (variant == V) {                                                                                                                     \
        hipLaunchKernelGGL((render_kernel<false, POOL, SCALAR, CULL, false, SPH>), g, t, lds_bytes, stream, P, img, acc, queue, none);    \
        return true;                                                                                                                        \
    }
    RT_VARIANT_TABLE(RT_LAUNCH)
#undef RT_LAUNCH
    return false;
}

// resident workgroups per CU of a variant at this dynamic-LDS size (advisory; an over-estimate only
// leaves late workgroups that find the queue empty)
int blocks_per_cu(unsigned variant, bool count, size_t lds_bytes, bool ext) {
    int n = 0;
    hipError_t e = hipErrorInvalidValue;
    if (count) {
#define RT_OCC_COUNT(V, SCALAR, CULL, EXT, SPH) \
    if (variant == V) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, render_kernel<true, true, SCALAR, CULL, EXT, SPH>, 256, lds_bytes);
        RT_COUNT_TABLE(RT_OCC_COUNT)
#undef RT_OCC_COUNT
    } else if (ext) {
#define RT_OCC_EXT(V, POOL, SCALAR, CULL, SPH) \
    if (variant == V) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, render_kernel<false, POOL, SCALAR, CULL, true, false>, 256, lds_bytes);
        RT_EXT_TABLE(RT_OCC_EXT)
#undef RT_OCC_EXT
    } else {
#define RT_OCC(V, POOL, SCALAR, CULL, SPH) \
    if (variant == V) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, render_kernel<false, POOL, SCALAR, CULL, false, SPH>, 256, lds_bytes);
        RT_VARIANT_TABLE(RT_OCC)
#undef RT_OCC
    }
    return (e == hipSuccess && n > 0) ? n : 4;
}

// candidate search of a variant (the CULL template argument): decides how much of the hot table is staged; -1: no such build
int variant_cull_mode(unsigned variant) {
#define RT_MODE(V, POOL, SCALAR, CULL, SPH) \
    if (variant == V) return CULL;
    RT_VARIANT_TABLE(RT_MODE)
#undef RT_MODE
    return -1;
}

// does the variant have a build with triangles and image textures?
bool variant_has_ext(unsigned variant) {
#define RT_HAS_EXT(V, POOL, SCALAR, CULL, SPH) \
    if (variant == V) return true;
    RT_EXT_TABLE(RT_HAS_EXT)
#undef RT_HAS_EXT
    return false;
}

bool variant_has_count(unsigned variant) {
#define RT_HAS_COUNT(V, SCALAR, CULL, EXT, SPH) \
    if (variant == V) return true;
    RT_COUNT_TABLE(RT_HAS_COUNT)
#undef RT_HAS_COUNT
    return false;
}

bool variant_exists(unsigned variant) { return variant == 0 || variant_cull_mode(variant) >= 0; }

__global__ void item_params_kernel(unsigned int *queue, ItemParams ip) {
    int *dst = reinterpret_cast<int *>(queue) + RT_ITEM_PARAMS_AT;
    // quad 0: tiles_x, bands, num_items, sample_first; quad 1: sample_count, spp_chunk, n_big, n_med;
    // quad 2: q_med, q_small, tile_rotate, -; quad 3: tile_rows, tile_first, tile_stride, local_rows
    const int v[16] = {ip.tiles_x, ip.bands, ip.num_items, ip.sample_first, ip.sample_count, ip.spp_chunk, ip.n_big,
                       ip.n_med, ip.q_med, ip.q_small, ip.tile_rotate, 0, ip.tile_rows, ip.tile_first, ip.tile_stride, ip.local_rows};
    for (int k = 0; k < 16; ++k) dst[k] = v[k];
}

void launch_item_params(unsigned int *queue, const ItemParams &ip, hipStream_t stream) {
    hipLaunchKernelGGL(item_params_kernel, dim3(1), dim3(1), 0, stream, queue, ip);
}

void launch_finalize(const unsigned long long *acc, float *out, size_t n, hipStream_t stream) {
    unsigned grid = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(finalize_kernel, dim3(grid), dim3(256), 0, stream, acc, out, n);
}

int set_max_dynamic_lds(size_t bytes) {
#define RT_ATTR1(K)                                                                                                  \
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&K), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return 1;
#define RT_ATTR(V, POOL, SCALAR, CULL, SPH) RT_ATTR1((render_kernel<false, POOL, SCALAR, CULL, false, SPH>))
    RT_VARIANT_TABLE(RT_ATTR)
#undef RT_ATTR
#define RT_ATTR_EXT(V, POOL, SCALAR, CULL, SPH) RT_ATTR1((render_kernel<false, POOL, SCALAR, CULL, true, false>))
    RT_EXT_TABLE(RT_ATTR_EXT)
#undef RT_ATTR_EXT
#define RT_ATTR_COUNT(V, SCALAR, CULL, EXT, SPH) RT_ATTR1((render_kernel<true, true, SCALAR, CULL, EXT, SPH>))
    RT_COUNT_TABLE(RT_ATTR_COUNT)
#undef RT_ATTR_COUNT
#undef RT_ATTR1
    return 0;
}
#endif  // RT_ISA_ONLY

}  // namespace rtmi
