/*
 * rt_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, fp32) of the reference's per-pixel path-tracing hot
 * path.  Nothing under ray-tracing-in-cuda_amd/ links, loads or calls this; it
 * is the checker for tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg.  See rt_oracle.c for the file:line map to the reference and
 * for how this restatement is pinned (oracle/_ref = the reference's own
 * cmake-cpu-version sources compiled with a hooked rand()).
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* record layouts deliberately equal include/rtmi.h's rt_prim / rt_material /
 * rt_texture so a test can pass the product's exported tables straight in
 * (tests assert the sizes match). */
typedef struct rto_prim {
    int32_t type;     /* 0 sphere, 1 xy_rect, 2 xz_rect, 3 yz_rect, 4 cylinder, 5 triangle (m = v1, v2, v3, unit
                         normal; m_inv[0..5] = texture coordinates of the three corners) */
    int32_t material;
    float f[6];
    float m[12];
    float m_inv[12];
} rto_prim;

typedef struct rto_material {
    int32_t type;     /* 0 lambertian, 1 metal, 2 dielectric, 3 diffuse_light */
    int32_t texture;
    float albedo[3];
    float fuzz;
    float ir;
} rto_material;

typedef struct rto_texture {
    int32_t type;     /* 0 solid, 1 checker (c0 even, c1 odd), 2 image (c0 = {image index, rows, cols}) */
    float c0[3];
    float c1[3];
} rto_texture;

typedef struct rto_image { /* taichi-version/material.py:96-110: rows x cols texels, R G B bytes */
    int32_t rows, cols;
    const uint8_t *rgb;
} rto_image;

typedef struct rto_camera_params {
    double lookfrom[3], lookat[3], vup[3];
    double vfov, aspect, aperture, focus_dist; /* aspect, focus_dist explicit */
} rto_camera_params;

typedef struct rto_camera {
    float origin[3], lower_left[3], horizontal[3], vertical[3];
    float u[3], v[3], w[3];
    float lens_radius;
} rto_camera;

#define RTO_FLAG_SKY_GRADIENT 1u
#define RTO_FLAG_DEFOCUS_BLUR 2u

typedef struct rto_scene {
    int32_t width, height, max_depth;
    uint32_t flags;
    float background[3];
    rto_camera cam;
    const rto_prim *prims;
    int32_t num_prims;
    const rto_material *mats;
    int32_t num_mats;
    const rto_texture *texs;
    int32_t num_texs;
    float rr_p; /* Russian-roulette survival probability per bounce, 0 = off */
    const rto_image *images; /* pixels of the image textures */
    int32_t num_images;
} rto_scene;

typedef struct rto_counts {
    uint64_t samples, queries, prim_tests, hits, misses;
    uint64_t scatter[4];
    uint64_t rng_draws;
} rto_counts;

void rto_derive_camera(const rto_camera_params *p, rto_camera *out);

/* radiance of one (pixel, sample); returns the number of hit queries */
int rto_sample(const rto_scene *s, uint64_t seed, int x, int y, int sample, float rgb[3],
               rto_counts *counts);

/* rows [y0, y1), samples [sample_first, sample_first+sample_count), summed in
 * chunks of spp_chunk (0 = one chunk); writes rgb_sum[(y*W+x)*3+c] of a full
 * W*H image buffer.  threads <= 0 -> all cores (OpenMP). counts may be NULL. */
int rto_trace_sample(const rto_scene *s, uint64_t seed, int x, int y, int sample, float rgb[3],
                     float *queries8, int max_queries);
int rto_render(const rto_scene *s, uint64_t seed, int y0, int y1, int sample_first,
               int sample_count, int spp_chunk, float *rgb_sum, rto_counts *counts, int threads);
/* the same over the pixel window [x0, x1) x [y0, y1) only (rgb_sum is still the full W*H buffer) */
int rto_render_rect(const rto_scene *s, uint64_t seed, int x0, int x1, int y0, int y1, int sample_first,
                    int sample_count, float *rgb_sum, rto_counts *counts, int threads);

/* bench.py's CPU-baseline timing of this restatement on a systematic row sample (seconds) */
double rto_time_sample(const rto_scene *s, uint64_t seed, int y0, int y1, int period, int band, int spp,
                       int threads, double *checksum, long long *pixels);

/* the fixed-sequence fp32 atan2 / acos behind the texture coordinates (known-answer tests) */
float rto_atan2f(float y, float x);
float rto_acosf(float c);
/* (u, v) of the hit record and the image-texture value there for the closest hit of the ray (o, d); returns 0 on a miss
 * (known-answer tests of the texture coordinates: gpu/object.cuh:87-93, 113-114, 283-288, taichi hittable.py:54-58, 233) */
int rto_hit_uv(const rto_scene *s, const float o[3], const float d[3], float uv[2], float *t, int *prim);

void rto_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void rto_sample_stream(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t *out, int n);
int rto_aabb_hit(const float bmin[3], const float bmax[3], const float orig[3],
                 const float dir[3], float t_min, float t_max);
/* PPM quantisation of one channel sum: int(256*clamp(sqrt(sum/spp),0,0.999)) */
int rto_quantize(float sum, int spp, int gamma);
int rto_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
