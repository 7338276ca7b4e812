// ref_harness.cpp -- TEST INFRASTRUCTURE ONLY; builds into oracle/_ref/ (git-ignored).
//
// Compiles the reference's OWN cmake-cpu-version sources (main.cpp and the
// headers it includes: ray_color, render, hittable_list::hit, sphere::hit,
// material::scatter, camera::get_ray ...) from where they lie under
// /root/reference -- nothing is copied into this repo; the Makefile passes the
// directory with -I -- and drives them deterministically:
//
//   * rand() is macro-hooked to a thread-local stream per (seed; pixel, sample)
//     (Philox4x32-10 block -> xorshift128), and RAND_MAX is redefined to 2^24-1, so the
//     reference's  rand() / double(int(RAND_MAX)+1)  (rtweekend.h:23) yields
//     exactly the 24-bit uniforms the HIP kernel and rt_oracle.c consume.
//     (The stock expression overflows with glibc's RAND_MAX = 2^31-1; the hook
//     also removes that portability bug.)
//   * must be built with clang++: `vec3(random_double(), random_double(), ...)`
//     (vec3.h:13,17,160) has unspecified argument evaluation order; clang
//     evaluates left to right (x first), which is the order this repo fixes.
//   * the scene is built by calling the reference constructors (sphere,
//     lambertian, metal, dielectric, solid_color, checker_texture, camera) with
//     the values of the scene tables under test.
//
// The reference's hittable_list holds std::vector<sphere> (hittable_list.h:8), so
// only sphere scenes with lambertian/metal/dielectric materials, sky-gradient
// background and defocus blur can be expressed -- configs 1, 3 and 5.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <iostream>
#include <limits>
#include <memory>
#include <random>
#include <sstream>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

// ---- the hook -----------------------------------------------------------------
namespace {
// the stream of one (pixel, sample): Philox4x32-10 block -> xorshift128 state (same as
// ray-tracing-in-cuda_amd/csrc/philox.h and rt_oracle.c; written out again here on purpose)
struct HookRng {
    uint32_t x, y, z, w;
    uint64_t draws;
};
thread_local HookRng g_rng = {0, 0, 0, 1, 0};

void hook_seed(uint64_t seed, uint32_t pixel, uint32_t sample) {
    uint32_t c0 = pixel, c1 = sample, c2 = 0, c3 = 0, k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
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
    g_rng.x = c0, g_rng.y = c1, g_rng.z = c2, g_rng.w = c3;
    if ((c0 | c1 | c2 | c3) == 0u) g_rng.w = 0x9E3779B9u;
    g_rng.draws = 0;
}
}  // namespace

int rt_ref_hook() {
    HookRng &g = g_rng;
    uint32_t t = g.x ^ (g.x << 11);
    g.x = g.y, g.y = g.z, g.z = g.w;
    g.w = g.w ^ (g.w >> 19) ^ (t ^ (t >> 8));
    g.draws++;
    return (int)(g.w >> 8);
}

#undef RAND_MAX
#define RAND_MAX 16777215
#define rand() rt_ref_hook()
#define main rt_ref_unused_main
#include "main.cpp"  // cmake-cpu-version/main.cpp, found through -I (see Makefile)
#undef main
#undef rand

// ---- C interface --------------------------------------------------------------
struct RefScene {
    hittable_list world;
    camera *cam = nullptr;
    std::vector<material *> mats;
    std::vector<texture *> texs;
};

extern "C" {

void *ref_scene_new(void) { return new RefScene(); }

void ref_scene_free(void *h) {
    RefScene *s = (RefScene *)h;
    if (!s) return;
    delete s->cam;
    // materials and textures are leaked on purpose: the reference's classes have no
    // virtual destructors (material.h:8-17, texture.hpp:6-9) and the reference itself
    // never frees them
    delete s;
}

// camera(lookfrom, lookat, vup, vfov, aspect_ratio, aperture, focus_dist), camera.h:9-16
void ref_scene_set_camera(void *h, const double lookfrom[3], const double lookat[3],
                          const double vup[3], double vfov, double aspect, double aperture,
                          double focus_dist) {
    RefScene *s = (RefScene *)h;
    delete s->cam;
    s->cam = new camera(point3(lookfrom[0], lookfrom[1], lookfrom[2]),
                        point3(lookat[0], lookat[1], lookat[2]), vec3(vup[0], vup[1], vup[2]), vfov,
                        aspect, aperture, focus_dist);
}

// mat_type: 0 lambertian (tex_type 0 solid c0 / 1 checker even=c0, odd=c1), 1 metal
// (albedo = c0, fuzz), 2 dielectric (ir).  Returns 0, or 1 for a material the
// reference's integrator cannot express.
int ref_scene_add_sphere(void *h, const double center[3], double radius, int mat_type, int tex_type,
                         const double c0[3], const double c1[3], double fuzz, double ir) {
    RefScene *s = (RefScene *)h;
    material *m = nullptr;
    if (mat_type == 0) {
        texture *t;
        if (tex_type == 0) t = new solid_color(color(c0[0], c0[1], c0[2]));
        else t = new checker_texture(color(c0[0], c0[1], c0[2]), color(c1[0], c1[1], c1[2]));
        s->texs.push_back(t);
        m = new lambertian(t);
    } else if (mat_type == 1) {
        m = new metal(color(c0[0], c0[1], c0[2]), fuzz);
    } else if (mat_type == 2) {
        m = new dielectric(ir);
    } else {
        return 1;
    }
    s->mats.push_back(m);
    sphere sp(point3(center[0], center[1], center[2]), radius, m);
    s->world.add(&sp);
    return 0;
}

int ref_scene_num_objects(void *h) { return (int)((RefScene *)h)->world.objects.size(); }

// one sample of one pixel through the reference's render() (main.cpp:45-55) with
// sample = 1, re-keying the stream first.  Returns the number of uniforms drawn.
int ref_sample(void *h, uint64_t seed, int x, int y, int sample, int width, int height, int max_depth,
               double rgb[3]) {
    RefScene *s = (RefScene *)h;
    hook_seed(seed, (uint32_t)(y * width + x), (uint32_t)sample);
    color c = render((double)x, (double)y, 1, *s->cam, s->world, max_depth, width, height);
    rgb[0] = c.x(), rgb[1] = c.y(), rgb[2] = c.z();
    return (int)g_rng.draws;
}

// rows [y0,y1) of the image, spp samples per pixel each on its own (pixel, sample)
// stream, summed in sample order in fp64 as render() does (main.cpp:47,52).
// rgb_sum is a full width*height*3 fp64 buffer, row 0 = bottom row.
int ref_render(void *h, uint64_t seed, int width, int height, int y0, int y1, int sample_first, int spp,
               int max_depth, double *rgb_sum, int threads) {
    RefScene *s = (RefScene *)h;
    if (!s || !s->cam || !rgb_sum) return 1;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#else
    threads = 1;
#endif
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
    for (int y = y0; y < y1; ++y) {
        for (int x = 0; x < width; ++x) {
            color res;
            for (int k = 0; k < spp; ++k) {
                hook_seed(seed, (uint32_t)(y * width + x), (uint32_t)(sample_first + k));
                res += render((double)x, (double)y, 1, *s->cam, s->world, max_depth, width, height);
            }
            double *o = rgb_sum + ((size_t)y * width + x) * 3;
            o[0] = res.x(), o[1] = res.y(), o[2] = res.z();
        }
    }
    return 0;
}

// CPU-baseline timing leg: the reference's pixel loop as it ships (main.cpp:99-106:
// one render(i, j, spp, cam, world, ...) call per pixel, which copies camera and
// world by value per pixel exactly as the reference does), rows spread over
// `threads` host threads.  One stream per pixel (the reference uses one global
// stream; thread safety needs at least per-thread streams).  Returns seconds.
double ref_time_rows(void *h, uint64_t seed, int width, int height, int y0, int y1, int spp,
                     int max_depth, int threads, double *checksum) {
    RefScene *s = (RefScene *)h;
    if (!s || !s->cam) return -1.0;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
    double t0 = omp_get_wtime();
#else
    threads = 1;
    clock_t c0 = clock();
#endif
    double acc = 0.0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads) reduction(+ : acc)
    for (int y = y0; y < y1; ++y) {
        for (int x = 0; x < width; ++x) {
            hook_seed(seed, (uint32_t)(y * width + x), 0u);
            color c = render((double)x, (double)y, spp, *s->cam, s->world, max_depth, width, height);
            acc += c.x() + c.y() + c.z();
        }
    }
    if (checksum) *checksum = acc;
#ifdef _OPENMP
    return omp_get_wtime() - t0;
#else
    return (double)(clock() - c0) / CLOCKS_PER_SEC;
#endif
}

// The CPU baseline's bounded sample: the reference's render() on rows y with ((y - y0) % period) < band,
// y in [y0, y1) -- every period-th band of `band` rows of the full-size frame.  One parallel loop over
// (row, 64-pixel span) so that every host core has work whatever the band height.
double ref_time_sample(void *h, uint64_t seed, int width, int height, int y0, int y1, int period, int band, int spp,
                       int max_depth, int threads, double *checksum, long long *pixels) {
    RefScene *s = (RefScene *)h;
    if (!s || !s->cam || period <= 0 || band <= 0) return -1.0;
    std::vector<int> rows;
    for (int y = y0; y < y1; ++y)
        if ((y - y0) % period < band) rows.push_back(y);
    const int spans = (width + 63) / 64;
    const long long jobs = (long long)rows.size() * spans;
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
        const int y = rows[(size_t)(j / spans)];
        const int xa = (int)(j % spans) * 64, xb = std::min(width, xa + 64);
        for (int x = xa; x < xb; ++x) {
            hook_seed(seed, (uint32_t)(y * width + x), 0u);
            color c = render((double)x, (double)y, spp, *s->cam, s->world, max_depth, width, height);
            acc += c.x() + c.y() + c.z();
        }
    }
    if (checksum) *checksum = acc;
    if (pixels) *pixels = (long long)rows.size() * width;
#ifdef _OPENMP
    return omp_get_wtime() - t0;
#else
    return (double)(clock() - c0) / CLOCKS_PER_SEC;
#endif
}

// the reference's write_color(out, c, spp) (color.h:14-35) for one pixel: the three
// integers it prints to the PPM
void ref_write_color(const double rgb_sum[3], int spp, int out[3]) {
    std::ostringstream os;
    write_color(os, color(rgb_sum[0], rgb_sum[1], rgb_sum[2]), spp);
    out[0] = out[1] = out[2] = -1;
    sscanf(os.str().c_str(), "%d %d %d", &out[0], &out[1], &out[2]);
}

int ref_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

}  // extern "C"
