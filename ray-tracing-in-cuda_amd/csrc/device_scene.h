// Device-side scene image shared between the host packer (render_hip.hip, host
// part) and the kernel.  One contiguous buffer, 16-byte records:
//
//   HOT part (copied into LDS by every workgroup; everything the primitive loop reads)
//     sphere  [ns + 4]  1 x float4   {cx, cy, cz, r*r}   big spheres first, then Morton-ordered
//                                clusters of 8; padded with never-hit records (r*r = -inf)
//     box     [ncl] 2 x float4   inflated bounding box of each cluster (culling variant)
//     rect    [nr]  2 x float4   {a0, a1, b0, b1} {k, axis(bits), 0, 0}
//     cyl     [nc]  6 x float4   m_inv rows 0..2, {radius^2, zmin, zmax, 0}, then the world-space bounding box of the open tube
//                                {min.xyz, _} {max.xyz, _}: one run of records, so that a lane that finds the cylinder in a cell's
//                                list has its box and its matrix in flight together
//     tri     [nt]  5 x float4   {v1.xyz, n.x} {v2.xyz, n.y} {v3.xyz, n.z}   (n = unit normal, hittable.py:104), then its box
//   COLD part (stays in global memory / L2; read once per bounce by the winning lane)
//     sphere  [ns]  1 x float4   {1/r, material(bits), list index(bits), material kind(bits)}
//     rect    [nr]  1 x float4   {material(bits), list index(bits), material kind(bits), 0}
//     cyl     [nc]  4 x float4   m rows 0..2, {material(bits), list index(bits), material kind(bits), 0}
//     tri     [nt]  2 x float4   {material(bits), list index(bits), u1.x, u1.y} {u2.x, u2.y, u3.x, u3.y}
//     mat     [nm]  3 x float4   {kind(bits), p0, p1, p2} {c0.xyz, p3} {c1.xyz, 0}
//     image   texels of the image textures, one 32-bit word each (r | g << 8 | b << 16), row-major
//
// Primitives are grouped by type (spheres, rects, cylinders), each group in list
// order; the original list index is kept for the reference's tie rule (a later
// object replaces an earlier one at equal t: hittable_list::hit accepts
// root <= closest_so_far, gpu-version/object.cuh:23-37 with :61).
#pragma once
#include <stdint.h>

// measurement variants, counting kernels and RTMI_* environment knobs (the default build); 0: the product kernels alone
#ifndef RTMI_ABLATIONS
#define RTMI_ABLATIONS 1
#endif

// spheres per cluster of the sphere table (a cluster occupies RT_CLUSTER + 1 slots, the last one never hit)
#define RT_CLUSTER 8

// records per cylinder / triangle of the hot tables: the primitive, then its bounding box (2 records)
#define RT_CYL_STRIDE 6
#define RT_TRI_STRIDE 5

// consecutive clusters under one outer box
#ifndef RT_GROUP
#define RT_GROUP 4
#endif

// slabs per axis of a window box in the range tables (candidate clusters of a ray segment): R[i0 * RT_SLABS + i1]
#ifndef RT_SLABS
#define RT_SLABS 16
#endif

// Pixel sums: signed 64-bit fixed point with RT_FIX_BITS fractional bits (include/rtmi.h, rt_render_hip_accumulate)
#define RT_FIX_BITS 24  /* = RT_ACC_FIX_BITS of include/rtmi.h (static_assert in render_kernel.hip) */
#define RT_FIX_CLAMP 65536.0f
#define RT_MAX_SAMPLES_PER_PIXEL (1 << 23)

namespace rtmi {

enum MatKind : int32_t {
    MK_LAMBERT_SOLID = 0,    // c0 = albedo
    MK_LAMBERT_CHECKER = 1,  // c0 = even, c1 = odd
    MK_LAMBERT_IMAGE = 2,    // c0 = {texel word offset (bits), rows (bits), cols (bits)}: image texture
    MK_METAL = 3,            // c0 = albedo, p0 = fuzz
    MK_DIELECTRIC = 4,       // p0 = ir, p1 = 1/ir, p2 = r0(1/ir), p3 = r0(ir)
    MK_LIGHT_SOLID = 5,      // c0 = emission
    MK_LIGHT_CHECKER = 6,    // c0 = even, c1 = odd
    MK_LIGHT_IMAGE = 7       // as MK_LAMBERT_IMAGE
};

// Per-launch values the kernel needs only when a wave fetches or flushes a work item.  They live in global
// memory behind the queue counter (16 ints at queue[RT_ITEM_PARAMS_AT], written by a one-thread kernel before the
// render launch; layout in item_params_kernel) and are read there, instead of occupying 14 SGPRs for the whole launch.
#define RT_ITEM_PARAMS_AT 16
struct ItemParams {
    int32_t tiles_x, bands, num_items, sample_first, sample_count, spp_chunk, n_big, n_med, q_med, q_small;
    int32_t tile_rows, tile_first, tile_stride, local_rows;
    int32_t tile_rotate;  // 1: the shard's k-th tile is k * tile_stride + ((tile_first - k) mod tile_stride) (include/rtmi.h)
};

// kernel parameter block (passed by value: lands in SGPRs / the kernarg segment)
struct RenderParams {
    int32_t off_cam;         // 6 records of the hot table: {origin, lens_radius}, lower_left, horizontal, vertical, u, v
    float background[3];
    uint32_t flags;
    float rr_p;              // Russian-roulette survival probability per bounce, 0 = off
    int32_t width, height, max_depth;
    float inv_wm1, inv_hm1;  // 1 / (W - 1), 1 / (H - 1) in fp32 (the jitter's scale, main.cu:96-97): computed by the host, because a
                             // value the kernel derives before its main loop is a VGPR that gets spilled to scratch
    // shard geometry (see rt_opts)
    int32_t tile_rows, tile_first, tile_stride, num_tiles, local_rows;
    // samples
    // samples [sample_first, +sample_count) are cut into three runs of chunks, handed out in this order by the
    // chunk-major queue: n_big chunks of spp_chunk, n_med chunks of q_med, then chunks of q_small to the end
    // (guided self-scheduling: the later an item is handed out, the shorter it is); num_chunks = total
    int32_t sample_first, sample_count, spp_chunk, num_chunks, n_big, n_med, q_med, q_small;
    int32_t orphan_max;      // an item is retired once at most this many of its paths are alive (render_kernel.hip)
    uint32_t seed_lo, seed_hi;
    // scene image
    int32_t ns, nr, nc, nm;
    int32_t nt;              // triangles (grouped ids ns + nr + nc ...)
    // the leading rectangles / cylinders / triangles of their tables that are tested for every query (the oversized ones: a
    // room's walls); the rest is listed in the cells of the wide grid tables.  The searches without a grid test all of them.
    int32_t nr_a, nc_a, nt_a;
    int32_t off_tri_hot, off_tri_cold;
    int32_t ns_pad;          // sphere slots incl. never-hit padding (= ns)
    int32_t np;              // leading slots that are always tested (big spheres), multiple of 8
    int32_t ncl;             // clusters of 8 slots after the prefix, each with a bounding box
    int32_t cluster;         // spheres per culling cluster (a multiple of 4, chosen per scene by the packer)
    int32_t off_box;         // 2 float4 per cluster: {min.xyz,_}, {max.xyz,_}
    int32_t ngr, off_gbox;   // outer boxes over RT_GROUP consecutive clusters
    int32_t nwin, off_wbox;  // window boxes over 64 consecutive clusters (64 / RT_GROUP outer boxes)
    // range tables: per window {box min.xyz}, {1 / slab width .xyz}, then per enabled axis RT_SLABS^2 64-bit masks
    int32_t off_rtab, rt_stride, rt_axes;  // float4 offset, float4 records per window, enabled axes (bit a)
    // uniform grid over the clustered spheres (CULL == 5): 4 header records, cells (one 32-bit word each: first item << 8 |
    // count), items (16-bit sphere slots); grid_cells == 0: the scene has no grid (no clustered spheres, or a cell with
    // more than 255 spheres)
    int32_t off_grid, off_grid_cells, off_grid_items, grid_cells;
    int32_t grid_wide;       // 1: wide grid tables (65536 sphere slots or more): 32-bit list entries, two 32-bit words per cell {first
                             // entry, (n_near << 8) | n_all}, up to 1023 cells per axis and 255 entries per cell (CULL == 7, global memory)
    int32_t grid_sheet;      // 1: the grid is one cell high (ny == 1): the walk steps along x and z only (CULL == 6)
    int32_t hot_vec4_grid;   // float4 count of the hot part through the grid tables (what the grid-walk kernel stages)
    int32_t hot_vec4_tables; // float4 count of the hot part including the range tables (what the range-table kernel stages)
    float cull_extent1;      // 1 + max |coordinate| of the clustered spheres (per-lane box margin, see packer)
    int32_t hot_vec4;        // float4 count of the hot part without the range tables (LDS bytes / 16 of the other variants)
    int32_t off_rect_hot;    // float4 offsets inside the image
    int32_t off_cyl_hot;
    int32_t off_sph_cold;
    int32_t off_rect_cold;
    int32_t off_cyl_cold;
    int32_t off_mat;
    int32_t tiles_x;         // 8-pixel tile columns
    int32_t bands;           // 8-row bands in the shard
    int32_t num_items;       // work items = tiles_x * bands * num_chunks (one wave each)
};

struct DevCounters {
    unsigned long long samples, queries, prim_tests, hits, misses;
    unsigned long long scatter[4];
    unsigned long long rng_draws;
    unsigned long long cand_lanes, cand_waves;  // sphere candidates resolved: per lane / per wave entry
    unsigned long long clusters_visited;         // culling: clusters whose spheres were tested (per wave)
    unsigned long long groups_visited;           // culling: outer boxes that passed (per wave)
    unsigned long long lane_clusters, lane_groups;  // culling: boxes that passed, per lane
    unsigned long long t_start_min, t_start_max, t_end_min, t_end_max;  // s_memrealtime (100 MHz) of wave starts / exits
    unsigned long long life_cycles, life_ticks;  // per-wave lifetime in shader cycles (s_memtime) and 100 MHz ticks, summed
    unsigned long long t_qe_min, t_qe_max;  // when a wave first found the queue empty
    unsigned int drain_hist[32];            // waves by time from queue-empty to exit, 50 us bins
    unsigned long long occ_hist[2][17];     // wave-queries by live lanes (bins of 4; 16 = all 64), [0] while the queue has items, [1] after
    unsigned int qe_hist[1024];             // waves by time from their start to queue-empty, 64 us bins
    unsigned int exit_hist[1024];           // waves by time from their start to exit, 64 us bins
    unsigned long long cycles[6];  // shader-clock time per main-loop section, summed over waves
    unsigned long long group_maxpop, query_maxpop;  // culling: max over lanes of needed clusters, per visited group / per wave-query
    unsigned long long wave_queries;             // closest-hit queries executed per wave (loop iterations)
    unsigned long long lane_cands;               // range tables: candidate clusters per lane (before the box test)
};

}  // namespace rtmi
