/*
 * rtmi.h -- C ABI of librtmi.so, the MI355X-native per-pixel path tracer.
 *
 * This is the drop-in boundary for the reference's hot path
 *     scene JSON  ->  render kernel  ->  float framebuffer  ->  PPM
 * Every entry point names the reference interface it replaces.  Paths are
 * relative to the reference checkout (gpu-version/ = CUDA renderer whose
 * interface is kept, cmake-cpu-version/ = CPU renderer whose ray_color
 * semantics are followed).
 *
 * Conventions
 *   - plain C: pointers, sizes, POD structs; no C++/torch/HIP types.
 *   - every function that can fail returns an rt_status (0 = RT_OK); the text
 *     of the last failure on the calling thread is at rt_last_error().
 *     The library never calls exit() (the reference does: rtweekend.cuh:41-53).
 *   - framebuffer layout is the reference's: rgb_sum[(y*W + x)*3 + c] holds the
 *     SUM over samples (not the mean), fp32, row y = 0 is the BOTTOM row
 *     (gpu-version/main.cu:76-78,102-104); division by spp and gamma are the
 *     writer's job (gpu-version/color.cuh:70-95).
 *   - there is no CPU fallback: rt_render_* fail with RT_ERR_HIP when no
 *     gfx950 device / runtime is usable.
 */
#ifndef RTMI_H
#define RTMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTMI_ABI_VERSION 3

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_ARG = 1,     /* null / out-of-range argument                      */
    RT_ERR_IO = 2,      /* file cannot be read / written                     */
    RT_ERR_JSON = 3,    /* malformed JSON text                               */
    RT_ERR_SCENE = 4,   /* well-formed JSON, invalid scene (unknown type...) */
    RT_ERR_HIP = 5,     /* HIP runtime / device error                        */
    RT_ERR_LIMIT = 6    /* an explicitly requested LDS-table kernel variant cannot hold the scene,
                           or the work-item count overflows (the default variant has no scene limit) */
} rt_status;

/* ---- table records (what the scene flattens to; also what the checker in
 *      tests/ reads back through rt_scene_get_*) ------------------------- */

/* gpu-version/rtweekend.cuh:70-91 (enum class class_type) */
typedef enum rt_prim_type {
    RT_PRIM_SPHERE = 0,   /* object.cuh:40-94   f = {cx,cy,cz,radius}        */
    RT_PRIM_XY_RECT = 1,  /* object.cuh:96-132  f = {x0,x1,y0,y1,k}          */
    RT_PRIM_XZ_RECT = 2,  /* object.cuh:134-164 f = {x0,x1,z0,z1,k}          */
    RT_PRIM_YZ_RECT = 3,  /* object.cuh:166-197 f = {y0,y1,z0,z1,k}          */
    RT_PRIM_CYLINDER = 4, /* object.cuh:216-297 f = {radius,zmin,zmax}, m/m_inv */
    RT_PRIM_TRIANGLE = 5  /* taichi-version/hittable.py:38-71, 95-110: m[0..8] = v1, v2, v3; m[9..11] = the unit
                             normal (v2-v1)x(v3-v1) / |..|; m_inv[0..5] = texture coordinates u1, u2, u3 (2 each) */
} rt_prim_type;

typedef enum rt_mat_type {
    RT_MAT_LAMBERTIAN = 0,    /* material.cuh:28-56   tex = albedo texture   */
    RT_MAT_METAL = 1,         /* material.cuh:58-75   albedo, fuzz (<=1)     */
    RT_MAT_DIELECTRIC = 2,    /* material.cuh:89-159  ir                     */
    RT_MAT_DIFFUSE_LIGHT = 3  /* material.cuh:161-182 tex = emission texture */
} rt_mat_type;

typedef enum rt_tex_type {
    RT_TEX_SOLID = 0,   /* texture.cuh:14-31  c0                             */
    RT_TEX_CHECKER = 1, /* texture.cuh:33-57  c0 = even, c1 = odd            */
    RT_TEX_IMAGE = 2    /* taichi-version/material.py:137-144: texel lookup by the hit record's (u, v);
                           c0 = {image index, rows, columns} as floats (exact: small integers) */
} rt_tex_type;

typedef struct rt_prim {
    int32_t type;      /* rt_prim_type                                       */
    int32_t material;  /* index into the material table                      */
    float f[6];        /* per-type parameters, see rt_prim_type              */
    float m[12];       /* cylinder object->world, rows of a 3x4 affine       */
    float m_inv[12];   /* cylinder world->object                             */
} rt_prim;

typedef struct rt_material {
    int32_t type;     /* rt_mat_type                                         */
    int32_t texture;  /* lambertian / diffuse_light: texture index, else -1  */
    float albedo[3];  /* metal                                               */
    float fuzz;       /* metal, clamped to <= 1 (material.cuh:61)            */
    float ir;         /* dielectric                                          */
} rt_material;

typedef struct rt_texture {
    int32_t type;  /* rt_tex_type */
    float c0[3];
    float c1[3];
} rt_texture;

/* camera.cuh:9-29 constructor arguments + the derived frame the kernel uses */
typedef struct rt_camera {
    float lookfrom[3], lookat[3], vup[3];
    float vfov;        /* degrees                                            */
    float aspect;      /* width / height                                     */
    float aperture;
    float focus_dist;
    /* derived (camera.cuh:18-28), computed in fp64 and rounded once          */
    float origin[3], lower_left[3], horizontal[3], vertical[3];
    float u[3], v[3], w[3];
    float lens_radius;
} rt_camera;

/* scene-level switches for the deltas between the reference's two renderers
 * (SURVEY.md appendix A). */
#define RT_FLAG_SKY_GRADIENT 1u /* miss colour = cmake-cpu-version/main.cpp:36-38
                                   lerp(white, (0.5,0.7,1)); else the JSON
                                   "background" constant (main.cu:63)         */
#define RT_FLAG_DEFOCUS_BLUR 2u /* camera.h:34 lens sampling on (the CUDA
                                   renderer has it commented out,
                                   camera.cuh:33-34)                          */

typedef struct rt_scene_info {
    int32_t width, height, samples_per_pixel, max_depth;
    int32_t num_prims, num_materials, num_textures;
    uint32_t flags;
    float background[3];
    float russian_roulette; /* survival probability per bounce, 0 = off (rt_scene_set_russian_roulette) */
} rt_scene_info;

typedef struct rt_scene rt_scene; /* opaque; gpu-version/parser.hpp:16-32 `struct scene` */

/* ---- scene I/O -------------------------------------------------------- */

/* parse_scene(filename), gpu-version/parser.hpp:504-573.  Same schema:
 * background[3] max_depth samples_per_pixel width height
 * camera{lookfrom lookat vup vfov aperture} object.data[] material.data[]
 * texture.data[] [output_file].  Extensions: texture type "checker"
 * {even[3], odd[3]} (texture.cuh:33-57 has the class, the parser lacks it), texture type "image" {"file": ppm} or
 * {"rows", "cols", "data": [r, g, b, ...]}, object types "triangle" {v1, v2, v3, [u1, u2, u3]} and "mesh" {"file":
 * obj, ["scale", "matrix"[9], "translate"]} (expanded into triangles when parsed),
 * optional top-level "sky_gradient": bool, "defocus_blur": bool (both default false: gpu-version renders a
 * constant background, main.cu:63, and has the lens sample disabled, camera.cuh:33-34) and
 * "russian_roulette": number in [0, 1].
 * Unknown object/material/texture "type" is a hard error (the reference
 * silently leaves the slot uninitialised). Returns NULL on failure. */
rt_scene *rt_scene_load_json(const char *path);
rt_scene *rt_scene_parse_json(const char *text, size_t len);

/* random_scene(), cmake-cpu-version/main.cpp:125-172 (and camera :89-94):
 * checker ground + 22x22 jittered small spheres + 3 big ones, sky gradient,
 * defocus blur, 16:9.  Randomness comes from Philox keyed by `seed`, drawn in
 * the order the reference draws (choose_mat, cx, cz, then material draws). */
rt_scene *rt_scene_rtiow(uint32_t seed, int width, int height, int spp, int max_depth);

/* serialise to the JSON schema above (round-trips through rt_scene_parse_json).
 * Returns bytes needed incl. NUL; writes at most cap bytes. */
size_t rt_scene_to_json(const rt_scene *s, char *out, size_t cap);

void rt_scene_free(rt_scene *s);

/* ---- programmatic scene building: the constructor argument lists of the
 *      reference classes (the C++ wrappers in rtmi.hpp call these) -------- */
rt_scene *rt_scene_new(int width, int height, int spp, int max_depth);
int rt_scene_set_background(rt_scene *s, const float rgb[3], uint32_t flags);
/* Russian roulette, the reference's 朴素光线追踪/4_0_path_tracing.py:43-46,88 (p_RR = 0.9): before every
 * closest-hit query the path survives with probability p (one extra draw; a path that does not
 * survive returns what it has collected so far) and a survivor's throughput is divided by p, so
 * the estimate stays unbiased (the reference divides after the scatter instead, which leaves the
 * last segment uncompensated: its images are darker by the factor p).  A non-parity fast mode: it changes which draws a sample
 * consumes, so images differ from p = 0 by noise, not bit for bit.  p = 0 (default) switches it off;
 * JSON: top-level "russian_roulette": p. */
int rt_scene_set_russian_roulette(rt_scene *s, float p);
/* camera(lookfrom, lookat, vup, vfov, aspect, aperture, focus_dist) camera.cuh:9-15;
 * aspect <= 0 -> width/height, focus_dist <= 0 -> |lookfrom-lookat| (parser.hpp:122-124) */
int rt_scene_set_camera(rt_scene *s, const float lookfrom[3], const float lookat[3],
                        const float vup[3], float vfov, float aspect, float aperture,
                        float focus_dist);
int rt_scene_add_solid_color(rt_scene *s, const float rgb[3]);                 /* -> texture id */
int rt_scene_add_checker(rt_scene *s, const float even[3], const float odd[3]); /* -> texture id */
int rt_scene_add_lambertian(rt_scene *s, int texture);                         /* -> material id */
int rt_scene_add_metal(rt_scene *s, const float albedo[3], float fuzz);
int rt_scene_add_dielectric(rt_scene *s, float ir);
int rt_scene_add_diffuse_light(rt_scene *s, int texture);
int rt_scene_add_sphere(rt_scene *s, const float center[3], float radius, int material); /* -> prim id */
/* axis: 0 = xy_rect (a=x,b=y,k=z), 1 = xz_rect, 2 = yz_rect */
int rt_scene_add_rect(rt_scene *s, int axis, float a0, float a1, float b0, float b1, float k,
                      int material);
/* cylinder(radius, zmin, zmax, mat) then rotate(axis, degrees) then translate(offset),
 * composed as parser.hpp:423-440 does (o2w = T * R); pass NULL to skip either. */
int rt_scene_add_cylinder(rt_scene *s, float radius, float zmin, float zmax, int material,
                          const float rot_axis[3], float rot_degrees, const float translate[3]);
/* negative ids above = -rt_status */

/* Image texture, taichi-version/material.py:96-110, 137-144 (the reference keeps one 100 x 100 image): `rgb` holds
 * rows x cols texels, 3 bytes each (R, G, B), row-major.  value(u, v, p) = texel[int(frac(u) * rows)][int(frac(v) *
 * cols)] / 255 (frac(x) = x - floor(x); an index that rounds up to rows / cols is clamped).  -> texture id.
 * The hit record's u, v (sphere object.cuh:87-93, rects :113-114, cylinder :283-288, triangle hittable.py:233)
 * are only evaluated for hits on materials with an image texture. */
int rt_scene_add_image_texture(rt_scene *s, int rows, int cols, const uint8_t *rgb);
/* the same from a file: PNG (8 bits per channel, non-interlaced; grey, grey + alpha, RGB, RGBA or palette; alpha is
 * dropped) or binary / text PPM (P6 / P3, maxval 255).  The reference reads its assets through OpenCV, and two of its
 * three "*.png" textures are JPEG files: convert those once
 * (`python -c "from PIL import Image; Image.open('bricks2.png').convert('RGB').save('bricks2.ppm')"`). */
int rt_scene_add_image_texture_file(rt_scene *s, const char *path);
/* rows / cols of image texture `texture` and, if out != NULL, its rows*cols*3 bytes; -rt_status on error */
int rt_scene_get_image(const rt_scene *s, int texture, int *rows, int *cols, uint8_t *out, size_t cap);
/* Triangle(v1, v2, v3, u1, u2, u3, material), taichi-version/hittable.py:95-110; uv pointers may be NULL (zeros) */
int rt_scene_add_triangle(rt_scene *s, const float v1[3], const float v2[3], const float v3[3],
                          const float uv1[2], const float uv2[2], const float uv3[2], int material);
/* readobj + the placement loop of taichi-version/main.py:23-41, 110-118: "v x y z", "vt u v", "f a b c" lines (1-based;
 * a face corner "a", "a/t" or "a/t/n"; without t the texture coordinate of corner a is vt[a], as in the reference);
 * every vertex is mapped to scale * (M v) + translate (M = 3x3 row-major, NULL = identity).  -> triangles added */
int rt_scene_add_obj(rt_scene *s, const char *path, int material, float scale, const float matrix[9],
                     const float translate[3]);

/* ---- animation (gpu-version/blue.py, blue2.py, dna.py: the frame harness) ----- */
/* blue.py:16-19 / blue2.py:16-19: add `degrees` to rotate.angle of every cylinder that has a
 * "rotate" and rebuild its transform; returns the number of cylinders changed (or -rt_status). */
int rt_scene_rotate_cylinders(rt_scene *s, double degrees);
int rt_scene_set_output_file(rt_scene *s, const char *path);
/* dna.py:17-98: the DNA frame at `angle_degrees` -- 60 emissive spheres + 30 emissive rotated
 * cylinders (3 helices x 10 rungs) placed into a copy of `base` (camera, background, size: the
 * reference uses basic_scene.json); base == NULL uses that file's values. */
rt_scene *rt_scene_dna(const rt_scene *base, double angle_degrees);
rt_scene *rt_scene_clone(const rt_scene *s);

/* CLI overrides -w -h -spp -d (cmake-cpu-version/main.cpp:71-81); <= 0 keeps the
 * value. Re-derives the camera when the aspect changes. */
int rt_scene_override(rt_scene *s, int width, int height, int spp, int max_depth);

/* ---- table read-back (host logic tests, checker input) ----------------- */
int rt_scene_get_info(const rt_scene *s, rt_scene_info *out);
int rt_scene_get_camera(const rt_scene *s, rt_camera *out);
int rt_scene_get_prims(const rt_scene *s, rt_prim *out, int cap);         /* -> count */
int rt_scene_get_materials(const rt_scene *s, rt_material *out, int cap); /* -> count */
int rt_scene_get_textures(const rt_scene *s, rt_texture *out, int cap);   /* -> count */

/* The device tables the host builds for a scene (no GPU needed: host logic tests, tools).  The reference rebuilds its object
 * graph on the device (move_to_device<<<1,1>>>, main.cu:374-446); here the host flattens the scene into ONE image of 16-byte
 * records -- sphere slots, the other primitives' records and boxes, the uniform grid over every primitive type (cells + lists),
 * cold records, materials -- which one memcpy uploads (csrc/device_scene.h, csrc/render_host.hip pack_scene). */
typedef struct rt_table_info {
    int32_t image_floats;      /* size of the image (rt_scene_table_image) */
    int32_t grid_wide;         /* 1: wide tables (32-bit entries, two words per cell: every primitive type listed); 0: compact
                                  (sphere-only scenes whose tables fit LDS: 16-bit entries, one word per cell) */
    int32_t grid_sheet;        /* compact tables, grid one cell high */
    int32_t grid_cells, grid_n[3];
    float grid_min[3], grid_size[3];
    float ob_near2, ob_far2;   /* squared reach of the lists' near / far tier (|ray origin|^2) */
    int32_t ns, np, ncl;       /* sphere slots, leading always-tested slots, clusters of 8 (+ 1 never-hit slot each) behind them */
    int32_t nr, nc, nt;        /* rectangles, cylinders, triangles ... */
    int32_t nr_a, nc_a, nt_a;  /* ... of which the leading ones are tested for every query instead of being listed */
    int32_t off_grid_cells, off_grid_items;                         /* record (float4) offsets into the image */
    int32_t off_sph_cold, off_rect_cold, off_cyl_cold, off_tri_cold; /* cold records: {.., material, list index, kind} */
    int32_t off_rect_hot, off_cyl_hot, off_tri_hot;
    int32_t hot_bytes_grid;    /* what a grid-walk kernel stages into LDS */
    int32_t kernel_variant;    /* what rt_opts.variant = 0 renders this scene with (2, 6, 16, 36 or 44) */
} rt_table_info;
int rt_scene_table_info(const rt_scene *s, rt_table_info *out);
/* copies min(cap_floats, image_floats) floats of the image; returns image_floats, or -rt_status */
int rt_scene_table_image(const rt_scene *s, float *out, int cap_floats);

/* ---- render ------------------------------------------------------------ */

typedef struct rt_opts {
    uint64_t seed;       /* Philox key that seeds every (pixel, sample) stream; the
                            reference seeds curand with the pixel id (main.cu:120-125) */
    int32_t device;      /* HIP device ordinal                                 */
    /* row-tile shard (multi-GPU): this call renders the row tiles
     *   t = tile_first, tile_first + tile_stride, ...   (< ceil(H / tile_rows))
     * tile t covers image rows [t*tile_rows, min(H,(t+1)*tile_rows)).
     * tile_stride <= 1 and tile_first == 0 -> whole image.                   */
    int32_t tile_rows;   /* 0 -> 8                                             */
    int32_t tile_first;
    int32_t tile_stride;
    /* How the tiles are dealt out to the tile_stride shards (N = tile_stride).  0: the plain interleave above.
     * 1: ROTATED interleave -- of every group of N consecutive tiles the shard tile_first owns one, and which one rotates
     *    from group to group: its k-th tile is k N + ((tile_first - k) mod N), i.e. tile t belongs to shard (t + t / N) mod N:
     *    no shard keeps one row phase of the image for itself.
     * 2: THERE AND BACK -- groups of 2 N tiles go to shards 0 .. N-1, then N-1 .. 0: the shard's tiles are 2 N j + tile_first and
     *    2 N j + 2 N - 1 - tile_first, which cancels the trend of the cost along the image inside every group.
     * All three give every shard the same number of tiles (+-1).  rt_shard_deal() says which one rt_render_hip_tiles and
     * bench.py use for a frame: 1 when the frame has >= 4 N^2 tiles (the rotation has gone round four times), else 2
     * (RTIOW 1080p, 135 tiles, busiest shard above the mean: N = 4: 0.36 % / 0.86 %, N = 8: 2.17 % / 0.92 % for 1 / 2). */
    int32_t tile_rotate;
    /* samples per work item (one wave renders an 8x8 tile x spp_chunk samples at a time).
     * Scheduling only: the per-pixel sum is exact (64-bit fixed point, 2^-24), so the
     * framebuffer does not depend on it.  0 -> 256 (less for small frames).          */
    int32_t spp_chunk;
    int32_t sample_first; /* render samples [sample_first, sample_first+count) */
    int32_t sample_count; /* 0 -> scene spp                                    */
    uint32_t variant;     /* 0 = the product kernel for this scene, one of
                               2  uniform-grid walk, compact tables in LDS, grid one cell high (walk along x and z: RTIOW)
                               6  uniform-grid walk, compact tables in LDS (sphere-only scenes that fit LDS)
                              36  uniform-grid walk over the wide tables in LDS: every primitive type is listed in the cells
                              44  the same with the tables in global memory (scenes too large for LDS: no size limit)
                              16  no culling: the reference's linear hittable_list scan (a handful of primitives of several
                                  types; also the definition the other kernels' images are held to)
                             (rt_stats.kernel_variant reports the choice).  Libraries built with RTMI_ABLATIONS (the default
                             build: rt_has_ablations()) also carry measurement variants with the same image, bit for bit:
                               1 = 6 with strict one-lane-per-pixel ownership, 40 = 6 with its tables in global memory,
                              17 = 16 with strict ownership, 24 = 16 with the tables in global memory, 32 = wave-level
                              cluster votes, 64 = per-lane cluster lists through a two-level box hierarchy, 128 = per-lane
                              cluster lists through range tables */
} rt_opts;

typedef struct rt_stats {
    double kernel_ms;     /* hipEvent time of the render launches of this call */
    double upload_ms;     /* scene table upload                                */
    int32_t launches;
    int32_t local_rows;   /* rows rendered by this shard                       */
    /* exact event counts, filled only by rt_render_hip_count */
    uint64_t samples, queries, prim_tests, hits, misses;
    uint64_t scatter[4];  /* per rt_mat_type */
    uint64_t rng_draws;
    uint64_t cand_lanes, cand_waves; /* sphere candidates that reached the sqrt block: lanes / wave entries */
    uint64_t clusters_visited;       /* culling: rounds of the per-lane cluster walk, per wave (= clusters a wave tested
                                        when it votes as a whole); grid: sphere-test passes per wave */
    uint64_t wave_queries;           /* closest-hit queries executed, counted per WAVE */
    uint64_t groups_visited;         /* culling: outer boxes that passed, per wave; range tables: rounds of the per-lane
                                        candidate box tests, per wave; grid: cell-step passes per wave */
    uint64_t lane_clusters;          /* culling: cluster boxes that passed, per LANE (what each ray needs) */
    uint64_t lane_groups;            /* culling: outer boxes that passed, per LANE; range tables: window boxes reached */
    uint64_t group_maxpop;           /* culling: max over lanes of needed clusters, summed over visited groups */
    uint64_t query_maxpop;           /* culling: max over lanes of needed clusters, summed over wave-queries; grid: LANES whose
                                        origin lies beyond the lists' reach and that scan every clustered sphere
                                        (group_maxpop: lanes that walk the far tier of the lists) */
    uint64_t cycles[6];              /* shader-clock time per main-loop section, summed over waves: refill, prefix
                                        spheres, culled spheres + rects + cylinders, shading, accumulation, loop control */
    int32_t cull_prefix, cull_clusters, cull_groups, cull_cluster_size; /* table geometry */
    double wave_start_spread_us, wave_end_spread_us, wave_span_us; /* first-to-last wave start / exit, first start
                                        to last exit (s_memrealtime) */
    /* rt_render_hip_tiles only: kernel_ms above is the SLOWEST device's render launches */
    uint64_t lane_cands;  /* culling by range tables: candidate clusters per LANE before the per-cluster box test;
                             grid: cell steps per LANE (lane_groups = lanes that entered the grid, lane_clusters =
                             sphere tests per LANE) */
    int32_t cull_mode;    /* candidate search of the kernel that ran: 5 uniform grid (6: its walk along x and z only, for a
                             grid one cell high), 3 range tables, 2 box hierarchy per lane, 1 wave votes, 0 none (flat scan) */
    int32_t cull_windows; /* windows of 64 clusters */
    double gather_ms;     /* root device: end of its own render -> assembled frame (ncclGather + row placement,
                             includes waiting for slower peers) */
    int32_t devices_used;
    int32_t grid_sheet;   /* rt_render_hip_count: 1 if the scene's grid is one cell high, i.e. the default kernel walks it along
                             x and z only (variant 2; the counting kernel itself walks in 3-D: same cells, same tests) */
    int32_t kernel_variant; /* the kernel that ran: what variant 0 (or a counting call) resolved to */
} rt_stats;

void rt_opts_default(rt_opts *o);

/* number of image rows / floats a shard owns (dense local buffer:
 * local row r <-> global row rt_shard_row(...)). */
int rt_shard_rows(const rt_scene *s, const rt_opts *o);
int rt_shard_global_row(const rt_scene *s, const rt_opts *o, int local_row);
/* the value of rt_opts.tile_rotate that rt_render_hip_tiles uses to cut this frame (o: tile_rows) into n_ranks shards
 * (blue.py:23-32 deals whole frames to GPUs; here the tiles of one frame) */
int rt_shard_deal(const rt_scene *s, const rt_opts *o, int n_ranks);

/* render<<<grid, 8x8>>>(spp, background, cam, world, max_depth, W, H, image, states)
 * gpu-version/main.cu:72-105 with its launch at :505-507.
 * d_rgb_sum: DEVICE pointer, rt_shard_rows()*W*3 floats, local rows dense.
 * stream: hipStream_t as void* (NULL = default stream). Asynchronous when
 * stats == NULL; with stats it records hipEvents and synchronises the stream.
 * One scene object keeps one set of device accumulators per device: overlapping
 * renders of the SAME rt_scene on one device must be issued on the same stream
 * (use rt_scene_clone for independent concurrent frames). */
int rt_render_hip_device(const rt_scene *s, const rt_opts *o, void *d_rgb_sum, void *stream,
                         rt_stats *stats);

/* same, host buffer in / out: allocates, launches, copies back, frees
 * (main.cu:482-513: cudaMallocManaged + render + cudaDeviceSynchronize). */
int rt_render_hip(const rt_scene *s, const rt_opts *o, float *rgb_sum, rt_stats *stats);

/* One frame on n_devices GPUs of this node (SURVEY.md 8(b) "Threading", 8(e); the reference's only multi-GPU
 * mechanism is one renderer process per GPU per animation FRAME, gpu-version/blue.py:23-32).  Row tile t
 * (o->tile_rows rows, default 8) is rendered by devices[t mod n] on a stream of its own -- the launch
 * rt_render_hip_device makes for that shard -- then ONE ncclGather (rccl.h:745, root = devices[0]) collects the
 * dense local buffers and a kernel on the root places the rows; rgb_sum (host, H*W*3 floats) receives the frame.
 * The result is bit-identical to rt_render_hip for every n.  devices == NULL means ordinals 0..n-1; the list must
 * not repeat a device.  o->device, tile_first and tile_stride are ignored (the call sets them per device).
 * Streams, RCCL communicators (ncclCommInitAll, ~0.1 s once per device list) and buffers are kept between calls;
 * rt_tiles_shutdown() releases them.  RCCL is loaded with dlopen at the first call: librtmi.so does not link it. */
int rt_render_hip_tiles(const rt_scene *s, const rt_opts *o, const int *devices, int n_devices,
                        float *rgb_sum, rt_stats *stats);
void rt_tiles_shutdown(void);

/* diagnostic launch of the same kernel with exact event counters (samples,
 * hit queries, primitive tests, ...) for the roofline's algorithmic flops.
 * rgb_sum may be NULL. */
int rt_render_hip_count(const rt_scene *s, const rt_opts *o, float *rgb_sum, rt_stats *stats);

/* Progressive / resumable rendering (the reference's only analogue is the running
 * average of the Taichi renderers, taichi-version/4_0_path_tracing.py).  `acc` is the caller's
 * exact pixel sums for this shard, [local_rows][width][3] signed 64-bit fixed point in units of
 * 2^-24 (zero-initialised before the first call; a sample is clamped to +-2^16 and at most 2^23 samples per
 * pixel are accepted, so the sums never wrap).  The call adds the samples
 * [sample_first, sample_first + sample_count) of `o` to it; integer addition is exact and
 * commutative, so any split of a sample range into calls, processes or devices gives the same
 * sums -- and the same framebuffer -- as one rt_render_hip call over the whole range.
 * If rgb_sum is not NULL it receives the fp32 framebuffer of the updated sums. */
int rt_render_hip_accumulate(const rt_scene *s, const rt_opts *o, int64_t *acc, float *rgb_sum,
                             rt_stats *stats);

/* fractional bits of the exact pixel sums (a file or buffer of sums written in another scale cannot be continued) */
#define RT_ACC_FIX_BITS 24
/* fp32 framebuffer values of exact sums: rgb_sum[i] = (float)(acc[i] * 2^-24), the conversion the
 * render path itself applies once per launch. */
void rt_acc_to_rgb(const int64_t *acc, float *rgb_sum, size_t n_values);

/* scatter a shard's dense local rows into a full-image buffer (host side of
 * the multi-GPU gather; also used after the RCCL gather on the root). */
int rt_shard_scatter_rows(const rt_scene *s, const rt_opts *o, const float *local_rgb,
                          float *full_rgb);

/* the same on the device, for a GATHERED buffer: d_gathered[n_ranks][pad_rows][W][3] holds rank r's dense local
 * rows (o->tile_rotate says how the shards were cut -- which rank owns tile t of the frame and as which of its
 * local tiles; pad_rows >= the largest shard), as
 * ncclGather delivers them; one kernel on `stream` (hipStream_t as void*) writes d_full[H][W][3].
 * rt_render_hip_tiles uses it on its root device. */
int rt_shard_place_rows_device(const rt_scene *s, const rt_opts *o, int n_ranks, int pad_rows,
                               const void *d_gathered, void *d_full, void *stream);

/* ---- output ------------------------------------------------------------ */

/* output_image(), gpu-version/main.cu:359-372 + write_color color.cuh:70-95:
 * "P3\n%d %d\n255\n" then "%d %d %d\n" per pixel, rows top to bottom,
 * value = int(256 * clamp(sqrt(sum/spp), 0, 0.999)). */
int rt_write_ppm(const char *path, const float *rgb_sum, int width, int height, int spp);
/* the same quantisation into a caller buffer of width*height*3 bytes, rows top
 * to bottom; gamma = 0 gives write_image()'s linear bytes (color.cuh:15-35). */
int rt_quantize_rgb8(const float *rgb_sum, int width, int height, int spp, int gamma,
                     uint8_t *out);

/* write_image(), gpu-version/color.cuh:15-35 (called at main.cu:514 with json["output_file"]):
 * 8-bit RGB PNG, rows top to bottom; gamma = 0 is the reference's linear image. */
int rt_write_png(const char *path, const float *rgb_sum, int width, int height, int spp, int gamma);
/* the scene's "output_file" (parser.hpp:566-567, default "main.png") */
const char *rt_scene_output_file(const rt_scene *s);

/* ---- misc -------------------------------------------------------------- */
const char *rt_last_error(void);
const char *rt_status_string(int status);
int rt_abi_version(void);
/* sizeof of the ABI structs as this library was compiled (binding self-checks):
 * 0 rt_opts, 1 rt_stats, 2 rt_prim, 3 rt_material, 4 rt_texture, 5 rt_camera, 6 rt_scene_info, 7 rt_table_info; else 0 */
size_t rt_struct_size(int which);
/* number of usable gfx950 devices, or -rt_status */
int rt_device_count(void);
/* 1 if this library carries the measurement variants of rt_opts.variant and the counting kernels of rt_render_hip_count
 * (the default build; `make ABLATIONS=0` builds the six product kernels alone: same ABI, rt_render_hip_count then
 * fails with RT_ERR_LIMIT and rt_opts.variant accepts 0, 2, 6, 16, 36, 44) */
int rt_has_ablations(void);
/* Philox4x32-10 block (seeds every (pixel, sample) stream), for known-answer tests */
void rt_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* first n raw 32-bit words of the xorshift128 stream of one (pixel, sample); the kernel's
 * uniforms are (word >> 8) * 2^-24 */
void rt_sample_stream(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t *out, int n);
/* slab test, gpu-version/aabb.hpp:15-29 (host evaluation of the device helper's formula) */
int rt_aabb_hit(const float bmin[3], const float bmax[3], const float orig[3], const float dir[3],
                float t_min, float t_max);

#ifdef __cplusplus
}
#endif
#endif /* RTMI_H */
