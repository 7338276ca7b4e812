// Host side of the render path: packs the scene tables into the device image
// (device_scene.h), keeps it resident per device, computes the shard geometry and
// launches render_kernel.  Replaces jsonmain()'s device set-up, gpu-version/main.cu:
// 462-513 (move_to_device<<<1,1>>>, cudaMallocManaged framebuffer, curand state
// allocation + init_random_library<<<W*H,1>>>, render<<<>>>, cudaDeviceSynchronize):
// one hipMemcpy of a few KB replaces the object-graph rebuild, and there is no RNG
// state to allocate or initialise.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>

#include "device_scene.h"
#include "scene.hpp"

#ifndef RT_WAVES_PER_SIMD
#define RT_WAVES_PER_SIMD 7
#endif
namespace rtmi {

bool launch_render(const RenderParams &P, const void *image, unsigned long long *acc, unsigned int *queue,
                   DevCounters *counters, size_t lds_bytes, unsigned grid, hipStream_t stream, unsigned variant, bool ext);
int blocks_per_cu(unsigned variant, bool count, size_t lds_bytes, bool ext);
bool variant_has_ext(unsigned variant);
bool variant_has_count(unsigned variant);
bool has_ablations();
void launch_finalize(const unsigned long long *acc, float *out, size_t n, hipStream_t stream);
void launch_item_params(unsigned int *queue, const ItemParams &ip, hipStream_t stream);
int set_max_dynamic_lds(size_t bytes);
bool variant_exists(unsigned variant);
int variant_cull_mode(unsigned variant);

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            set_error("HIP error %d (%s) at %s:%d: %s", (int)e_, hipGetErrorString(e_), __FILE__, __LINE__, #expr); \
            return RT_ERR_HIP;                                                                \
        }                                                                                     \
    } while (0)

struct DeviceEntry {
    int device = -1;
    uint64_t version = 0;
    void *d_image = nullptr;
    size_t image_bytes = 0;
    unsigned long long *d_acc = nullptr;  // fixed-point pixel accumulators of the last launch
    size_t acc_bytes = 0;
    DevCounters *d_counters = nullptr;
    float *d_out = nullptr;  // framebuffer of the host-buffer entry points (rt_render_hip), kept between calls
    size_t out_bytes = 0;
    int num_cus = 0;
};

struct DeviceSceneCache {
    std::mutex mu;  // guards the packed image and the entry list -- not the launches of an entry
    std::vector<std::unique_ptr<DeviceEntry>> entries;  // stable addresses: one entry per device
    // host-side packed image (rebuilt when the scene version changes)
    uint64_t packed_version = 0;
    std::vector<float> image;  // float4 records
    RenderParams layout;       // ns/nr/nc/nm + offsets filled by pack
    ~DeviceSceneCache() {
        int cur = 0;
        bool have = hipGetDevice(&cur) == hipSuccess;
        for (auto &ep : entries) {
            DeviceEntry &e = *ep;
            if (e.device < 0) continue;
            if (hipSetDevice(e.device) != hipSuccess) continue;
            if (e.d_image) (void)hipFree(e.d_image);
            if (e.d_acc) (void)hipFree(e.d_acc);
            if (e.d_counters) (void)hipFree(e.d_counters);
            if (e.d_out) (void)hipFree(e.d_out);
        }
        if (have) (void)hipSetDevice(cur);
    }
};

static inline float bits(int32_t v) {
    float f;
    memcpy(&f, &v, 4);
    return f;
}

// LDS left for a workgroup's hot tables beside full occupancy (RT_WAVES_PER_SIMD workgroups per CU), after one 64-pixel
// rgb accumulator per wave
static constexpr size_t kAccLds = 4 * 192 * sizeof(unsigned long long);
static constexpr size_t kLdsTableBytes = (size_t)(160 * 1024 / RT_WAVES_PER_SIMD) - kAccLds;

#if RTMI_ABLATIONS
// measurement knobs of the default build (tests, bench.py, tools/): read once; the product build (make ABLATIONS=0) has none
static double knob(const char *name, double fallback) {
    const char *e = getenv(name);
    return e ? atof(e) : fallback;
}
static bool knob_set(const char *name) { return getenv(name) != nullptr; }
#else
static constexpr double knob(const char *, double fallback) { return fallback; }
static constexpr bool knob_set(const char *) { return false; }
#endif

// ---- scene tables -> device image ------------------------------------------------
// forced: primitives that must be tested for every query whatever their size (members of a cell whose list overflowed in an
// earlier attempt).  Returns false, with more primitives added to `forced`, when a cell's list overflows.
static bool pack_scene_once(const Scene &s, DeviceSceneCache &c, std::vector<char> &forced) {
    std::vector<int> sph, rec, cyl, tri;
    for (size_t i = 0; i < s.prims.size(); ++i) {
        switch (s.prims[i].type) {
        case RT_PRIM_SPHERE: sph.push_back((int)i); break;
        case RT_PRIM_CYLINDER: cyl.push_back((int)i); break;
        case RT_PRIM_TRIANGLE: tri.push_back((int)i); break;
        default: rec.push_back((int)i); break;
        }
    }
    const bool sphere_only = rec.empty() && cyl.empty() && tri.empty() && s.images.empty();
    // Sphere slots.  The closest hit does not depend on the visiting order (ties are resolved
    // through the stored list index), so the table is laid out for the kernel:
    //   prefix   : the big spheres (|r| > 4 x median), largest first -- the likeliest closest
    //              hits, tested unconditionally, so best_t is tight before anything else;
    //   clusters : the rest in Morton order of their centres, 8 per cluster (what the grid's cells list; the clusters
    //              and their boxes serve the cluster searches of the ablation builds and the scan of far origins).
    // Both parts are padded with never-hit records (r*r = -inf).
    std::stable_sort(sph.begin(), sph.end(), [&](int a, int b) {
        return std::fabs(s.prims[a].f[3]) > std::fabs(s.prims[b].f[3]);
    });
    std::vector<int> slots;  // prim index per slot, -1 = padding
    std::vector<int> rest;
    {
        float big = 0.0f;  // 0: every sphere is tested for every query (16 spheres or fewer)
        if (sph.size() > 16) {
            std::vector<float> radii;
            for (int i : sph) radii.push_back(std::fabs(s.prims[i].f[3]));
            std::nth_element(radii.begin(), radii.begin() + radii.size() / 2, radii.end());
            big = 4.0f * radii[radii.size() / 2];
        }
        for (int i : sph) {
            if (big == 0.0f || std::fabs(s.prims[i].f[3]) > big || forced[i]) slots.push_back(i);
            else rest.push_back(i);
        }
    }
    while (slots.size() % 4) slots.push_back(-1);  // the prefix is walked four records at a time
    const int np_slots = (int)slots.size();
    if (!rest.empty()) {
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int i : rest)
            for (int a = 0; a < 3; ++a) lo[a] = std::min(lo[a], s.prims[i].f[a]), hi[a] = std::max(hi[a], s.prims[i].f[a]);
        auto spread = [](uint32_t v) {  // 10 bits -> every third bit
            v = (v | (v << 16)) & 0x030000FFu;
            v = (v | (v << 8)) & 0x0300F00Fu;
            v = (v | (v << 4)) & 0x030C30C3u;
            v = (v | (v << 2)) & 0x09249249u;
            return v;
        };
        auto morton = [&](int i) {
            uint32_t q[3];
            for (int a = 0; a < 3; ++a) {
                float ext = hi[a] - lo[a];
                float t = ext > 0 ? (s.prims[i].f[a] - lo[a]) / ext : 0.0f;
                q[a] = (uint32_t)std::min(1023.0f, std::max(0.0f, t * 1023.0f));
            }
            return spread(q[0]) | (spread(q[1]) << 1) | (spread(q[2]) << 2);
        };
        std::stable_sort(rest.begin(), rest.end(), [&](int a, int b) { return morton(a) < morton(b); });
    }
    // Cluster q occupies the slots [np + 9 q, + 8) followed by ONE never-hit slot: with a stride of 9 records, record h of
    // clusters q and q' lies (q - q') records apart modulo 16, so the lanes of a wave that read different clusters hit
    // different LDS banks with the same instruction (a ds_read_b128 serves 16 lanes per cycle, one 16-byte record per 4
    // banks; with stride 16 every cluster's record h shared one bank group: 19.5 % of the LDS cycles were conflicts).
    // All-padding clusters end the table (read-ahead of the flat scan; the pair test's never-hit partner).
    const int csize = RT_CLUSTER;
    const int n_clusters = ((int)rest.size() + csize - 1) / csize;
    const int cstride = csize + 1;
    for (int q = 0; q < n_clusters + 3; ++q)  // + 3 all-padding clusters: the flat scan reads 16 records a step and one ahead
        for (int h = 0; h < cstride; ++h) {
            const size_t j = (size_t)q * csize + h;
            slots.push_back((q < n_clusters && h < csize && j < rest.size()) ? rest[j] : -1);
        }
    while (slots.size() % 4) slots.push_back(-1);
    const int ns_slots = (int)slots.size();

    // ---- world boxes of the other primitives (double precision; grown below where they are listed)
    struct OBox {
        double lo[3], hi[3];
    };
    auto rect_box = [&](const rt_prim &p) {
        OBox b;
        const int axis = p.type - RT_PRIM_XY_RECT;  // 0: z = k (x, y extents), 1: y = k (x, z), 2: x = k (y, z)
        const int ia = axis == 2 ? 1 : 0, ib = axis == 0 ? 1 : 2, ik = axis == 0 ? 2 : (axis == 1 ? 1 : 0);
        b.lo[ia] = std::min(p.f[0], p.f[1]), b.hi[ia] = std::max(p.f[0], p.f[1]);
        b.lo[ib] = std::min(p.f[2], p.f[3]), b.hi[ib] = std::max(p.f[2], p.f[3]);
        b.lo[ik] = b.hi[ik] = p.f[4];
        return b;
    };
    // cylinders: world box of the open tube = union of the boxes of its two end circles
    // (centre M (0,0,z), radius R, normal = the tube axis a: half-extent R sqrt(1 - a_i^2) on axis i)
    auto cyl_box = [&](const rt_prim &p, double R, double zpad) {
        OBox b;
        double ax[3] = {p.m[2], p.m[6], p.m[10]};  // image of the object z axis
        const double an = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
        const double z0 = std::min((double)p.f[1], (double)p.f[2]) - zpad, z1 = std::max((double)p.f[1], (double)p.f[2]) + zpad;
        for (int a = 0; a < 3; ++a) {
            const double ai = an > 0 ? ax[a] / an : 0.0;
            const double half = R * std::sqrt(std::max(0.0, 1.0 - ai * ai));
            const double c0 = p.m[a * 4 + 2] * z0 + p.m[a * 4 + 3];
            const double c1 = p.m[a * 4 + 2] * z1 + p.m[a * 4 + 3];
            b.lo[a] = std::min(c0, c1) - half, b.hi[a] = std::max(c0, c1) + half;
        }
        return b;
    };
    auto tri_box = [&](const rt_prim &p) {
        OBox b;
        for (int a = 0; a < 3; ++a) {
            b.lo[a] = std::min((double)p.m[a], std::min((double)p.m[3 + a], (double)p.m[6 + a]));
            b.hi[a] = std::max((double)p.m[a], std::max((double)p.m[3 + a], (double)p.m[6 + a]));
        }
        return b;
    };
    std::vector<int> others;  // rects, cylinders, triangles: prim indices
    others.insert(others.end(), rec.begin(), rec.end());
    others.insert(others.end(), cyl.begin(), cyl.end());
    others.insert(others.end(), tri.begin(), tri.end());
    std::vector<int> oidx(s.prims.size(), -1);  // position of a primitive in `others`
    for (size_t k = 0; k < others.size(); ++k) oidx[others[k]] = (int)k;
    std::vector<OBox> obox(others.size());
    for (size_t k = 0; k < others.size(); ++k) {
        const rt_prim &p = s.prims[others[k]];
        obox[k] = p.type == RT_PRIM_CYLINDER ? cyl_box(p, std::fabs((double)p.f[0]), 0.0) : (p.type == RT_PRIM_TRIANGLE ? tri_box(p) : rect_box(p));
    }
    // Which of them go into the grid's cells?  Like the spheres: none while the scene is small (16 primitives outside the
    // sphere prefix or fewer: the per-query loops are the cheaper search), and not the oversized ones (largest box edge > 8 x
    // the median of what would be listed: a room's walls, a ground plane), which every ray has to test anyway.
    std::vector<char> listed(others.size(), 0);
    if (rest.size() + others.size() > 16) {
        std::vector<double> sizes;
        for (int i : rest) sizes.push_back(2.0 * std::fabs((double)s.prims[i].f[3]));
        auto edge = [&](size_t k) {
            return std::max(obox[k].hi[0] - obox[k].lo[0], std::max(obox[k].hi[1] - obox[k].lo[1], obox[k].hi[2] - obox[k].lo[2]));
        };
        for (size_t k = 0; k < others.size(); ++k) sizes.push_back(edge(k));
        std::nth_element(sizes.begin(), sizes.begin() + sizes.size() / 2, sizes.end());
        const double big = 8.0 * sizes[sizes.size() / 2];
        for (size_t k = 0; k < others.size(); ++k) listed[k] = (edge(k) <= big || !(big > 0.0)) && !forced[others[k]];
    }
    // the other primitives' tables: the always-tested ones first (the kernel's per-query loops run over that prefix)
    auto order_table = [&](std::vector<int> &v, int &n_always) {
        std::vector<int> a, b;
        for (int i : v) (listed[oidx[i]] ? b : a).push_back(i);
        n_always = (int)a.size();
        v = a;
        v.insert(v.end(), b.begin(), b.end());
    };
    RenderParams &L = c.layout;
    memset(&L, 0, sizeof L);
    {
        int na = 0;
        order_table(rec, na), L.nr_a = na;
        order_table(cyl, na), L.nc_a = na;
        order_table(tri, na), L.nt_a = na;
    }
    L.ns = ns_slots, L.nr = (int)rec.size(), L.nc = (int)cyl.size(), L.nm = (int)s.mats.size();
    L.nt = (int)tri.size();
    L.ns_pad = ns_slots;
    L.np = np_slots;
    L.ncl = n_clusters;
    L.cluster = csize;
    // grouped id of an other primitive: its position in the reordered tables behind the sphere slots
    std::vector<int> gid_of(s.prims.size(), -1);
    for (size_t k = 0; k < rec.size(); ++k) gid_of[rec[k]] = ns_slots + (int)k;
    for (size_t k = 0; k < cyl.size(); ++k) gid_of[cyl[k]] = ns_slots + L.nr + (int)k;
    for (size_t k = 0; k < tri.size(); ++k) gid_of[tri[k]] = ns_slots + L.nr + L.nc + (int)k;

    int off = 0;
    off += ns_slots + 4;  // sphere hot (+ never-hit padding)
    const int n_groups = (n_clusters + RT_GROUP - 1) / RT_GROUP;  // RT_GROUP consecutive clusters share an outer box
    L.ngr = n_groups;
    const int groups_per_window = 64 / RT_GROUP;  // one 64-bit cluster mask per window in the kernel
    const int n_windows = (n_groups + groups_per_window - 1) / groups_per_window;
    L.nwin = n_windows;
    L.off_rect_hot = off;
    off += 2 * L.nr;
    L.off_cyl_hot = off;
    off += RT_CYL_STRIDE * L.nc;  // (each followed by its box)
    L.off_tri_hot = off;
    off += RT_TRI_STRIDE * L.nt;
    L.off_cam = off;  // camera::camera's derived vectors (camera.h:9-31): read once per new sample
    off += 6;
    // Range tables (ablation variant 128: candidate clusters of a ray segment without testing every box): per window of 64
    // clusters and per enabled axis, R[i0 * 16 + i1] = the clusters whose box overlaps the slabs i0..i1 of the window box cut
    // into RT_SLABS slabs along that axis (64-bit mask).  An axis along which the clustered spheres do not spread (a sheet:
    // RTIOW's y) carries no information and is left out (2 KB of LDS).
    int axes = 0;
    {
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int i : rest) {
            const float r = std::fabs(s.prims[i].f[3]);
            for (int a = 0; a < 3; ++a) lo[a] = std::min(lo[a], s.prims[i].f[a] - r), hi[a] = std::max(hi[a], s.prims[i].f[a] + r);
        }
        float ext[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
        const float big = std::max(ext[0], std::max(ext[1], ext[2]));
        for (int a = 0; a < 3; ++a)
            if (!rest.empty() && ext[a] > 0.05f * big) axes |= 1 << a;
    }
    int n_axes = (axes & 1) + ((axes >> 1) & 1) + ((axes >> 2) & 1);
    // ---- uniform grid (the candidate search: every lane walks the cells its ray crosses front to back -- 3-D DDA -- and tests
    // what they list).  A primitive is listed in every cell its GROWN box touches.  The growth covers the fp32 error of its
    // test, so that the walk finds every hit the linear scan would find:
    //   spheres: |disc_fp32 - disc| <= K eps a |oc|^2 (K = 32 bounds the operation-by-operation sum, about 15 eps |oc|^2), so a
    //     ray the fp32 test can accept passes within r' = sqrt(r^2 + K eps |oc|^2) of the centre, and its fp32 hit point lies
    //     inside that ball too.  |oc| <= |o| + |c|, so the growth depends on how far from the coordinate origin a ray starts;
    //     the lists come in two tiers:
    //       near  |o| <= ob_near (the cloud, the camera; RTIOW 23.4): the first n_near entries of a cell's list
    //       far   |o| <= ob_far  (hits on distant ground; 8x the cloud, at least 64): all n_all entries
    //   cylinders (object.cuh:233-290): the same quadratic in the tube's object space, K = 64 (the transform's rounding rides
    //     along): tube radius R' = sqrt(R^2 + K eps (ob_far + |corner|)^2), ends moved out by the term below;
    //   rectangles, triangles: the accepted point lies on the ray within a few eps (|o| + |p|) of the primitive's plane (the
    //     triangle's plane point r = o - d/|d| (oc.n)/theta carries the error of oc.n, which does not grow with 1/theta) and, in
    //     projection, inside its outline to the same order: 64 eps (ob_far + |corner|);
    //   one tier (the far one) for these three: their growth is 1e-4 of a cell.
    // Lanes further out than ob_far test the grid's bounds with the per-lane margin of the box tests and, if they can reach it
    // at all, test everything the cells list: rare, and the flat scan is the definition of the result.
    // (0.004 cell + 1e-5 (max|c| + 1)) more covers the walk's own rounding: the entry point, the cell boundaries, up to
    // 1023 accumulated leave distances.)
    // Two table formats: COMPACT (sphere-only scenes that fit LDS: 16-bit entries, one word per cell, <= 255 cells per axis,
    // <= 63 entries per cell) and WIDE (everything else: 32-bit entries, two words per cell, <= 1023 cells per axis, <= 1023
    // sphere entries per tier and <= 4095 other entries per cell).
    std::vector<uint32_t> grid_cells;   // compact: (first item << 12) | (n_near << 6) | n_all;  wide: {first item, n_near | n_all << 10 | n_other << 20}
    std::vector<uint32_t> grid_items;   // sphere slots (a cell's near-tier entries first), then grouped ids of the other primitives
    bool grid_wide = !sphere_only || ns_slots >= 65536 || knob_set("RTMI_FORCE_WIDE");  // (the knob: measurement)
    float grid_min[3] = {0, 0, 0}, grid_size[3] = {1, 1, 1};
    int grid_n[3] = {0, 0, 0};
    float grid_ob2[2] = {0.0f, 0.0f}, grid_shrink = 0.0f;
    std::vector<OBox> listed_box(others.size());  // grown boxes of the listed others (also what their box tests read)
    size_t n_listed = rest.size();
    for (size_t k = 0; k < others.size(); ++k) n_listed += listed[k] ? 1 : 0;
    bool overflow = false;
    for (int attempt = 0; attempt < 2 && n_listed > 0; ++attempt) {
        const double cell_factor = knob("RTMI_GRID_CELL", 1.0);
        const double ob_env = knob("RTMI_GRID_OB", 0.0);  // experiments
        grid_cells.clear(), grid_items.clear();
        const double max_dim = grid_wide ? 1023.0 : 255.0;
        const long long max_cells = grid_wide ? (1LL << 21) : (1LL << 18);
        const size_t max_per_cell = grid_wide ? 1023 : 63, max_other = 4095;
        const size_t max_items = grid_wide ? ((size_t)1 << 30) : ((size_t)1 << 20);
        // centres (spheres) and box centres (others): the cloud the cells are sized for
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300}, cmax = 0.0, cmax2 = 0.0;
        double slo[3] = {1e300, 1e300, 1e300}, shi[3] = {-1e300, -1e300, -1e300};  // of the sphere centres alone
        auto add_point = [&](const double *c) {
            double c2 = 0.0;
            for (int a = 0; a < 3; ++a) {
                lo[a] = std::min(lo[a], c[a]), hi[a] = std::max(hi[a], c[a]);
                cmax = std::max(cmax, std::fabs(c[a])), c2 += c[a] * c[a];
            }
            cmax2 = std::max(cmax2, std::sqrt(c2));
        };
        for (int i : rest) {
            const double c[3] = {s.prims[i].f[0], s.prims[i].f[1], s.prims[i].f[2]};
            add_point(c);
            for (int a = 0; a < 3; ++a) slo[a] = std::min(slo[a], c[a]), shi[a] = std::max(shi[a], c[a]);
        }
        for (size_t k = 0; k < others.size(); ++k) {
            if (!listed[k]) continue;
            const double c[3] = {0.5 * (obox[k].lo[0] + obox[k].hi[0]), 0.5 * (obox[k].lo[1] + obox[k].hi[1]), 0.5 * (obox[k].lo[2] + obox[k].hi[2])};
            add_point(c);
            // (the far corners count towards the reach of the tiers: |oc| <= |o| + |corner|)
            double far2 = 0.0;
            for (int a = 0; a < 3; ++a) {
                const double m = std::max(std::fabs(obox[k].lo[a]), std::fabs(obox[k].hi[a]));
                far2 += m * m, cmax = std::max(cmax, m);
            }
            cmax2 = std::max(cmax2, std::sqrt(far2));
        }
        const double cam = std::sqrt(s.cam.lookfrom[0] * s.cam.lookfrom[0] + s.cam.lookfrom[1] * s.cam.lookfrom[1] +
                                     s.cam.lookfrom[2] * s.cam.lookfrom[2]) + std::fabs(s.cam.aperture);
        const double ob_near = ob_env > 0.0 ? ob_env : std::max(1.5 * cmax2, 1.1 * cam + 1.0);
        const double ob_far = std::max(std::max(64.0, 8.0 * cmax2), 4.0 * ob_near);
        grid_ob2[0] = (float)(ob_near * ob_near * (1.0 - 1e-5)), grid_ob2[1] = (float)(ob_far * ob_far * (1.0 - 1e-5));
        double ext[3], big = 0.0;
        for (int a = 0; a < 3; ++a) ext[a] = hi[a] - lo[a], big = std::max(big, ext[a]);
        int dims = 0;
        double measure = 1.0;
        bool spread[3];
        for (int a = 0; a < 3; ++a) {
            spread[a] = ext[a] > 0.05 * big;
            if (spread[a]) ++dims, measure *= ext[a];
        }
        // cell edge: a multiple of the spacing of the centres.  Measured: RTIOW (a sheet, one sphere per unit square)
        // 1.0 / 1.25 / 1.5 / 2.0 x -> 41.5 / 39.9 / 42.0 / 42.0 ms per 256 spp; 4000 spheres in a volume 0.7 / 1.0 /
        // 1.4 x -> 6.4 / 6.7 / 7.3 ms, 20000: 12.3 / 12.7 / 15.1 ms (RTMI_GRID_CELL scales the choice).
        double cell = dims ? std::pow(measure / (double)n_listed, 1.0 / dims) * (dims == 3 ? 0.85 : 1.25) * cell_factor : 1.0;
        if (!(cell > 0.0)) cell = 1.0;
        std::vector<double> grow_near(rest.size()), grow_far(rest.size());
        double rmax_near = 0.0, rmax_far = 0.0;
        double blo[3], bhi[3], nlo[3], nhi[3];  // bounds of the far-tier boxes (the grid's), of the near-tier boxes
        const double eps = std::ldexp(1.0, -24);
        for (;;) {
            rmax_near = rmax_far = 0.0;
            const double walk = 4e-3 * cell + 1e-5 * (cmax + 1.0);
            for (size_t k = 0; k < rest.size(); ++k) {
                const float *sp = s.prims[rest[k]].f;
                const double r = std::fabs((double)sp[3]);
                const double cn = std::sqrt((double)sp[0] * sp[0] + (double)sp[1] * sp[1] + (double)sp[2] * sp[2]);
                const double K = 32.0 * eps;
                grow_near[k] = std::sqrt(r * r + K * (ob_near + cn) * (ob_near + cn)) + walk;
                grow_far[k] = std::sqrt(r * r + K * (ob_far + cn) * (ob_far + cn)) + walk;
                rmax_near = std::max(rmax_near, grow_near[k]), rmax_far = std::max(rmax_far, grow_far[k]);
            }
            for (int a = 0; a < 3; ++a) {  // (empty without spheres: slo = +huge, shi = -huge)
                blo[a] = slo[a] - rmax_far, bhi[a] = shi[a] + rmax_far;
                nlo[a] = slo[a] - rmax_near, nhi[a] = shi[a] + rmax_near;
            }
            for (size_t k = 0; k < others.size(); ++k) {
                if (!listed[k]) continue;
                const rt_prim &p = s.prims[others[k]];
                double corner2 = 0.0;
                for (int a = 0; a < 3; ++a) {
                    const double m = std::max(std::fabs(obox[k].lo[a]), std::fabs(obox[k].hi[a]));
                    corner2 += m * m;
                }
                const double reach = ob_far + std::sqrt(corner2);
                const double g = 64.0 * eps * reach + walk;
                OBox b = obox[k];
                if (p.type == RT_PRIM_CYLINDER) {
                    const double R = std::fabs((double)p.f[0]);
                    b = cyl_box(p, std::sqrt(R * R + 64.0 * eps * reach * reach), 64.0 * eps * reach);
                }
                for (int a = 0; a < 3; ++a) {
                    b.lo[a] -= g, b.hi[a] += g;
                    blo[a] = std::min(blo[a], b.lo[a]), bhi[a] = std::max(bhi[a], b.hi[a]);
                    nlo[a] = std::min(nlo[a], b.lo[a]), nhi[a] = std::max(nhi[a], b.hi[a]);
                }
                listed_box[k] = b;
            }
            long long total = 1;
            for (int a = 0; a < 3; ++a) {
                const double span = bhi[a] - blo[a];
                grid_n[a] = spread[a] ? (int)std::min(max_dim, std::max(1.0, std::ceil(span / cell))) : 1;
                grid_min[a] = (float)blo[a];
                grid_size[a] = (float)(span / grid_n[a]);
                total *= grid_n[a];
            }
            if (total <= max_cells) break;
            cell *= 1.3;
        }
        // near-tier lanes clip their rays to the bounds of the near-tier boxes: the far tier's, this much further in
        double shrink = 1e300;
        for (int a = 0; a < 3; ++a) shrink = std::min(shrink, std::min(nlo[a] - blo[a], bhi[a] - nhi[a]));
        grid_shrink = (float)(std::max(0.0, shrink) * (1.0 - 1e-6));
        const int nx = grid_n[0], ny = grid_n[1], nz = grid_n[2];
        std::vector<std::vector<uint32_t>> lists((size_t)nx * ny * nz), extra((size_t)nx * ny * nz), olist((size_t)nx * ny * nz);
        auto cell_of = [&](int a, double x) {
            const int i = (int)std::floor((x - (double)grid_min[a]) / (double)grid_size[a]);
            return std::min(std::max(i, 0), grid_n[a] - 1);
        };
        for (size_t k = 0; k < rest.size(); ++k) {
            // the slot of this sphere: clusters of csize behind the prefix, one padding slot per cluster
            const int slot = np_slots + (int)(k / csize) * cstride + (int)(k % csize);
            int c0[3], c1[3], n0[3], n1[3];
            for (int a = 0; a < 3; ++a) {
                const double c = (double)s.prims[rest[k]].f[a];
                c0[a] = cell_of(a, c - grow_far[k]), c1[a] = cell_of(a, c + grow_far[k]);
                n0[a] = cell_of(a, c - grow_near[k]), n1[a] = cell_of(a, c + grow_near[k]);
            }
            for (int iz = c0[2]; iz <= c1[2]; ++iz)
                for (int iy = c0[1]; iy <= c1[1]; ++iy)
                    for (int ix = c0[0]; ix <= c1[0]; ++ix) {
                        const bool near = ix >= n0[0] && ix <= n1[0] && iy >= n0[1] && iy <= n1[1] && iz >= n0[2] && iz <= n1[2];
                        (near ? lists : extra)[((size_t)iz * ny + iy) * nx + ix].push_back((uint32_t)slot);
                    }
        }
        // the other primitives; one that would be listed in more than 4096 cells is tested for every query instead
        std::vector<char> too_wide(others.size(), 0);
        for (size_t k = 0; k < others.size(); ++k) {
            if (!listed[k]) continue;
            int c0[3], c1[3];
            long long cells = 1;
            for (int a = 0; a < 3; ++a) {
                c0[a] = cell_of(a, listed_box[k].lo[a]), c1[a] = cell_of(a, listed_box[k].hi[a]);
                cells *= c1[a] - c0[a] + 1;
            }
            if (cells > 4096) {
                too_wide[k] = 1;
                continue;
            }
            for (int iz = c0[2]; iz <= c1[2]; ++iz)
                for (int iy = c0[1]; iy <= c1[1]; ++iy)
                    for (int ix = c0[0]; ix <= c1[0]; ++ix) olist[((size_t)iz * ny + iy) * nx + ix].push_back((uint32_t)gid_of[others[k]]);
        }
        bool retry = false;
        for (size_t k = 0; k < others.size(); ++k)
            if (too_wide[k]) forced[others[k]] = 1, retry = true;
        if (retry) return false;
        grid_cells.resize(lists.size() * (grid_wide ? 2 : 1));
        overflow = false;
        for (size_t cidx = 0; cidx < lists.size(); ++cidx) {
            const size_t n_near = lists[cidx].size(), n_all = n_near + extra[cidx].size(), n_other = olist[cidx].size();
            if (n_all > max_per_cell || n_other > max_other || grid_items.size() + n_all + n_other >= max_items) {
                overflow = true;
                if (grid_wide) {  // a clump even for the wide tables: its members are tested for every query from now on
                    for (uint32_t slot : lists[cidx]) forced[slots[slot]] = 1;
                    for (uint32_t slot : extra[cidx]) forced[slots[slot]] = 1;
                    for (uint32_t g : olist[cidx]) {
                        const int k = (int)g - ns_slots;  // position in the reordered tables: rects, cylinders, triangles
                        forced[k < L.nr ? rec[k] : (k < L.nr + L.nc ? cyl[k - L.nr] : tri[k - L.nr - L.nc])] = 1;
                    }
                }
                continue;
            }
            if (grid_wide)
                grid_cells[2 * cidx] = (uint32_t)grid_items.size(),
                                 grid_cells[2 * cidx + 1] = (uint32_t)n_near | ((uint32_t)n_all << 10) | ((uint32_t)n_other << 20);
            else
                grid_cells[cidx] = ((uint32_t)grid_items.size() << 12) | ((uint32_t)n_near << 6) | (uint32_t)n_all;
            grid_items.insert(grid_items.end(), lists[cidx].begin(), lists[cidx].end());
            grid_items.insert(grid_items.end(), extra[cidx].begin(), extra[cidx].end());
            grid_items.insert(grid_items.end(), olist[cidx].begin(), olist[cidx].end());
        }
        if (overflow && grid_wide) return false;
        // compact tables must also leave the kernel its full occupancy: otherwise the wide ones, read from global memory
        if (!grid_wide) {
            const size_t hot = (size_t)(off + 4 + ((int)grid_cells.size() + 3) / 4 + ((int)grid_items.size() + 1 + 7) / 8) * 16;
            if (overflow || hot > (size_t)knob("RTMI_GLOBAL_TABLE_BYTES", (double)kLdsTableBytes)) {
                grid_wide = true;
                continue;  // once more, in the wide format
            }
        }
        break;
    }
    if (n_listed == 0) grid_cells.clear(), grid_items.clear(), grid_n[0] = grid_n[1] = grid_n[2] = 0;
    L.grid_cells = (int)(grid_cells.size() / (grid_wide ? 2 : 1));
    L.grid_wide = grid_wide ? 1 : 0;
    L.grid_sheet = (!grid_cells.empty() && grid_n[1] == 1 && !grid_wide) ? 1 : 0;
    L.off_grid = off;  // 4 records {min.xyz, ob_near^2} {1/size.xyz, ob_far^2} {size.xyz, shrink} {nx, ny, nz, -}, then cells, then items
    off += 4;
    L.off_grid_cells = off;
    off += ((int)grid_cells.size() + 3) / 4;
    L.off_grid_items = off;
    off += grid_wide ? ((int)grid_items.size() + 1 + 3) / 4 : ((int)grid_items.size() + 1 + 7) / 8;  // (+ 1: the pair test reads one entry past a list)
    L.hot_vec4_grid = off;  // what the grid-walk kernels stage into LDS
    // the boxes of the cluster searches (ablation builds) lie behind the grid tables, so that the grid walk does not stage
    // them (RTIOW: 2.5 KB of 15.2 KB)
    L.off_box = off;
    off += 2 * n_clusters;
    L.off_wbox = off;
    off += 2 * n_windows;
    L.off_gbox = off;  // outer boxes: the box-hierarchy variant reads them
    off += 2 * n_groups;
    L.hot_vec4 = off;  // what the box-hierarchy and flat-scan variants stage into LDS
    L.rt_axes = axes;
    L.rt_stride = 2 + n_axes * (RT_SLABS * RT_SLABS / 2);  // float4 records per window: {min, 1/width} + masks (2 per record)
    L.off_rtab = off;
    off += n_windows * L.rt_stride;
    L.hot_vec4_tables = off;  // ... and the range-table kernel: the same plus the tables
    L.off_sph_cold = off;
    off += ns_slots;
    L.off_rect_cold = off;
    off += L.nr;
    L.off_cyl_cold = off;
    off += 4 * L.nc;
    L.off_tri_cold = off;
    off += 2 * L.nt;
    L.off_mat = off;
    off += 3 * L.nm;
    // texels of the image textures: one 32-bit word each, every image starts on a float4 record
    std::vector<int> image_word(s.images.size(), 0);
    for (size_t k = 0; k < s.images.size(); ++k) {
        image_word[k] = off * 4;
        off += (int)(((size_t)s.images[k].rows * s.images[k].cols + 3) / 4);
    }
    c.image.assign((size_t)(off > 0 ? off : 1) * 4, 0.0f);
    float *I = c.image.data();
    auto rec4 = [&](int idx) { return I + (size_t)idx * 4; };

    {
        rt_camera cam;
        derive_camera(s, &cam);
        const float *src[6] = {cam.origin, cam.lower_left, cam.horizontal, cam.vertical, cam.u, cam.v};
        for (int k = 0; k < 6; ++k) {
            float *h = rec4(L.off_cam + k);
            h[0] = src[k][0], h[1] = src[k][1], h[2] = src[k][2];
        }
        rec4(L.off_cam)[3] = cam.lens_radius;
    }
    for (int k = 0; k < ns_slots + 4; ++k) {
        float *h = rec4(k);
        const int pi = k < ns_slots ? slots[k] : -1;
        if (pi < 0) {
            h[3] = -INFINITY;  // c = +inf, disc = -inf: never a candidate
            continue;
        }
        const rt_prim &p = s.prims[pi];
        h[0] = p.f[0], h[1] = p.f[1], h[2] = p.f[2];
        h[3] = p.f[3] * p.f[3];  // r*r in fp32, as sphere::hit evaluates it
        float *cd = rec4(L.off_sph_cold + k);
        cd[0] = 1.0f / p.f[3];   // (p - c) / r  ==  (1/r) * (p - c), vec3.cuh:105
        cd[1] = bits(p.material);
        cd[2] = bits(pi);
    }
    // Boxes tested per lane with a margin (the cluster boxes of the ablation searches; the boxes of the cylinders and
    // triangles that are tested for every query).  Skipping a box must never change the result of the fp32 test behind
    // it, whose rounding error grows with the distance |oc| from the ray origin: with unit roundoff e = 2^-24,
    // |disc_fp32 - disc| <= 15 e a |oc|^2, so a ray the sphere test can accept passes within r + sqrt(15 e)|oc| ~ r + 1e-3 |oc|
    // of the centre, and its fp32 root lies within the same distance of that approach point: the hit point is inside the
    // sphere's box grown by 2e-3 |oc|.  |oc| <= sqrt(3) (max|o_i| + extent), so the KERNEL grows every such box per lane by
    //     m = 4e-3 (max|o_i| + extent + 1)
    // (two shifted ray origins per query, no extra work per box); a ray that leaked 2000 units inside the ground sphere
    // thereby visits everything, exactly like the noise it would hit.  The stored boxes only carry a 1e-5-relative pad
    // for their own rounding (the listed cylinders and triangles: their grid growth, which is larger).
    float extent = 0.0f;  // max |coordinate| reached by a clustered sphere, a cylinder or a triangle
    for (int k = np_slots; k < (int)slots.size(); ++k) {
        if (slots[k] < 0) continue;
        const rt_prim &p = s.prims[slots[k]];
        for (int a = 0; a < 3; ++a) extent = std::max(extent, std::fabs(p.f[a]) + std::fabs(p.f[3]));
    }
    auto other_index = [&](int prim) { return (size_t)oidx[prim]; };
    for (int k = 0; k < L.nc; ++k) {
        const OBox &b = obox[other_index(cyl[k])];
        for (int a = 0; a < 3; ++a) extent = std::max(extent, (float)std::max(std::fabs(b.lo[a]), std::fabs(b.hi[a])));
    }
    for (int k = 0; k < L.nt; ++k) {
        const rt_prim &p = s.prims[tri[k]];
        for (int cc = 0; cc < 9; ++cc) extent = std::max(extent, std::fabs(p.m[cc]));
    }
    L.cull_extent1 = extent + 1.0f;
    const float inflate = 1e-5f * (extent + 1.0f);
    auto store_box = [&](float *b, int prim) {
        const size_t k = other_index(prim);
        const bool grown = listed[k] && !grid_cells.empty();
        const OBox &src = grown ? listed_box[k] : obox[k];
        for (int a = 0; a < 3; ++a) {
            // (rounded outwards: the grown box is a double-precision bound)
            b[a] = std::nextafterf((float)src.lo[a], -INFINITY) - inflate;
            b[4 + a] = std::nextafterf((float)src.hi[a], INFINITY) + inflate;
        }
    };
    for (int k = 0; k < L.nt; ++k) store_box(rec4(L.off_tri_hot + RT_TRI_STRIDE * k + 3), tri[k]);
    for (int k = 0; k < L.nc; ++k) store_box(rec4(L.off_cyl_hot + RT_CYL_STRIDE * k + 4), cyl[k]);
    for (int q = 0; q < n_clusters; ++q) {
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int k = 0; k < csize; ++k) {
            const int pi = slots[np_slots + cstride * q + k];
            if (pi < 0) continue;
            const rt_prim &p = s.prims[pi];
            const float r = std::fabs(p.f[3]);
            for (int a = 0; a < 3; ++a) lo[a] = std::min(lo[a], p.f[a] - r), hi[a] = std::max(hi[a], p.f[a] + r);
        }
        float *b = rec4(L.off_box + 2 * q);
        for (int a = 0; a < 3; ++a) {
            b[a] = lo[a] - inflate;
            b[4 + a] = hi[a] + inflate;
        }
    }
    for (int g = 0; g < n_groups; ++g) {  // outer boxes: union of the (already inflated) cluster boxes
        float *gb = rec4(L.off_gbox + 2 * g);
        for (int a = 0; a < 3; ++a) gb[a] = INFINITY, gb[4 + a] = -INFINITY;
        for (int q = g * RT_GROUP; q < std::min(n_clusters, (g + 1) * RT_GROUP); ++q) {
            const float *b = rec4(L.off_box + 2 * q);
            for (int a = 0; a < 3; ++a) gb[a] = std::min(gb[a], b[a]), gb[4 + a] = std::max(gb[4 + a], b[4 + a]);
        }
    }
    for (int w = 0; w < n_windows; ++w) {  // third level (big scenes): union of the window's outer boxes
        float *wb = rec4(L.off_wbox + 2 * w);
        for (int a = 0; a < 3; ++a) wb[a] = INFINITY, wb[4 + a] = -INFINITY;
        for (int g = w * groups_per_window; g < std::min(n_groups, (w + 1) * groups_per_window); ++g) {
            const float *b = rec4(L.off_gbox + 2 * g);
            for (int a = 0; a < 3; ++a) wb[a] = std::min(wb[a], b[a]), wb[4 + a] = std::max(wb[4 + a], b[4 + a]);
        }
    }
    // range tables of every window (layout: see L.off_rtab above)
    for (int w = 0; w < n_windows; ++w) {
        const float *wb = rec4(L.off_wbox + 2 * w);
        float *hd = rec4(L.off_rtab + w * L.rt_stride);
        uint64_t *masks = reinterpret_cast<uint64_t *>(hd + 8);
        const int q0 = w * 64, q1 = std::min(n_clusters, q0 + 64);
        int ai = 0;
        for (int a = 0; a < 3; ++a) {
            const float lo = wb[a], hi = wb[4 + a];
            const float width = (hi - lo) / (float)RT_SLABS;
            hd[a] = lo;
            hd[4 + a] = width > 0.0f ? 1.0f / width : 0.0f;  // a window that is flat on this axis: every point -> slab 0
            if (!((axes >> a) & 1)) continue;
            uint64_t slab[RT_SLABS];
            // the kernel finds a point's slab as floor((x - lo) * (1 / width)) in fp32: grow every slab by a tolerance
            // far above that rounding so that a cluster touching a slab boundary is listed on both sides
            const float tol = 1e-3f * width + 1e-5f * (std::fabs(lo) + std::fabs(hi));
            for (int i = 0; i < RT_SLABS; ++i) {
                const float a0 = lo + width * (float)i - tol, a1 = lo + width * (float)(i + 1) + tol;
                uint64_t m = 0;
                for (int q = q0; q < q1; ++q) {
                    const float *b = rec4(L.off_box + 2 * q);
                    // the first and last slab also stand for everything outside the window box on their side
                    const bool over = (i == 0 || b[4 + a] >= a0) && (i == RT_SLABS - 1 || b[a] <= a1);
                    if (over || !(width > 0.0f)) m |= 1ull << (q - q0);
                }
                slab[i] = m;
            }
            uint64_t *R = masks + (size_t)ai * RT_SLABS * RT_SLABS;
            for (int i0 = 0; i0 < RT_SLABS; ++i0) {
                uint64_t m = 0;
                for (int i1 = 0; i1 < RT_SLABS; ++i1) {
                    if (i1 >= i0) m |= slab[i1];
                    R[i0 * RT_SLABS + i1] = i1 >= i0 ? m : 0;
                }
            }
            ++ai;
        }
    }
    {  // grid tables
        float *g = rec4(L.off_grid);
        for (int a = 0; a < 3; ++a) {
            g[a] = grid_min[a];
            g[4 + a] = grid_size[a] > 0.0f ? 1.0f / grid_size[a] : 0.0f;
            g[8 + a] = grid_size[a];
            g[12 + a] = bits(grid_n[a]);
        }
        g[3] = grid_ob2[0], g[7] = grid_ob2[1], g[11] = grid_shrink;
        if (!grid_cells.empty()) memcpy(rec4(L.off_grid_cells), grid_cells.data(), grid_cells.size() * sizeof(uint32_t));
        if (!grid_items.empty()) {
            if (grid_wide) {
                memcpy(rec4(L.off_grid_items), grid_items.data(), grid_items.size() * sizeof(uint32_t));
            } else {
                uint16_t *dst = reinterpret_cast<uint16_t *>(rec4(L.off_grid_items));
                for (size_t i = 0; i < grid_items.size(); ++i) dst[i] = (uint16_t)grid_items[i];
            }
        }
    }
    for (int k = 0; k < L.nr; ++k) {
        const rt_prim &p = s.prims[rec[k]];
        float *h = rec4(L.off_rect_hot + 2 * k);
        h[0] = p.f[0], h[1] = p.f[1], h[2] = p.f[2], h[3] = p.f[3];
        h[4] = p.f[4];
        h[5] = bits(p.type - RT_PRIM_XY_RECT);
        float *cd = rec4(L.off_rect_cold + k);
        cd[0] = bits(p.material);
        cd[1] = bits(rec[k]);
    }
    for (int k = 0; k < L.nc; ++k) {
        const rt_prim &p = s.prims[cyl[k]];
        float *h = rec4(L.off_cyl_hot + RT_CYL_STRIDE * k);
        memcpy(h, p.m_inv, 12 * sizeof(float));
        h[12] = p.f[0] * p.f[0], h[13] = p.f[1], h[14] = p.f[2];
        float *cd = rec4(L.off_cyl_cold + 4 * k);
        memcpy(cd, p.m, 12 * sizeof(float));
        cd[12] = bits(p.material);
        cd[13] = bits(cyl[k]);
    }
    for (int k = 0; k < L.nt; ++k) {
        const rt_prim &p = s.prims[tri[k]];
        float *h = rec4(L.off_tri_hot + RT_TRI_STRIDE * k);
        for (int cc = 0; cc < 3; ++cc) {
            h[4 * cc] = p.m[3 * cc], h[4 * cc + 1] = p.m[3 * cc + 1], h[4 * cc + 2] = p.m[3 * cc + 2];
            h[4 * cc + 3] = p.m[9 + cc];
        }
        float *cd = rec4(L.off_tri_cold + 2 * k);
        cd[0] = bits(p.material), cd[1] = bits(tri[k]);
        cd[2] = p.m_inv[0], cd[3] = p.m_inv[1];
        cd[4] = p.m_inv[2], cd[5] = p.m_inv[3], cd[6] = p.m_inv[4], cd[7] = p.m_inv[5];
    }
    for (size_t k = 0; k < s.images.size(); ++k) {
        const SceneImage &im = s.images[k];
        uint32_t *w = reinterpret_cast<uint32_t *>(I) + image_word[k];
        for (size_t t = 0; t < (size_t)im.rows * im.cols; ++t)
            w[t] = (uint32_t)im.rgb[3 * t] | ((uint32_t)im.rgb[3 * t + 1] << 8) | ((uint32_t)im.rgb[3 * t + 2] << 16);
    }
    for (int k = 0; k < L.nm; ++k) {
        const rt_material &m = s.mats[k];
        float *q = rec4(L.off_mat + 3 * k);
        int kind = MK_LAMBERT_SOLID;
        const rt_texture *t = (m.texture >= 0 && m.texture < (int)s.texs.size()) ? &s.texs[m.texture] : nullptr;
        switch (m.type) {
        case RT_MAT_LAMBERTIAN:
        case RT_MAT_DIFFUSE_LIGHT: {
            bool light = m.type == RT_MAT_DIFFUSE_LIGHT;
            bool checker = t && t->type == RT_TEX_CHECKER;
            kind = light ? (checker ? MK_LIGHT_CHECKER : MK_LIGHT_SOLID) : (checker ? MK_LAMBERT_CHECKER : MK_LAMBERT_SOLID);
            if (t && t->type == RT_TEX_IMAGE) {
                kind = light ? MK_LIGHT_IMAGE : MK_LAMBERT_IMAGE;
                const size_t im = (size_t)t->c0[0];
                q[4] = bits(image_word[im]), q[5] = bits(s.images[im].rows), q[6] = bits(s.images[im].cols);
            } else if (t) {
                q[4] = t->c0[0], q[5] = t->c0[1], q[6] = t->c0[2];
                q[8] = t->c1[0], q[9] = t->c1[1], q[10] = t->c1[2];
            }
            break;
        }
        case RT_MAT_METAL:
            kind = MK_METAL;
            q[1] = m.fuzz;
            q[4] = m.albedo[0], q[5] = m.albedo[1], q[6] = m.albedo[2];
            break;
        case RT_MAT_DIELECTRIC: {
            kind = MK_DIELECTRIC;
            float ir = m.ir, inv_ir = 1.0f / m.ir;
            // reflectance()'s r0 for both refraction ratios, material.cuh:175-178
            float r0f = (1.0f - inv_ir) / (1.0f + inv_ir);
            r0f = r0f * r0f;
            float r0b = (1.0f - ir) / (1.0f + ir);
            r0b = r0b * r0b;
            q[1] = ir, q[2] = inv_ir, q[3] = r0f, q[7] = r0b;
            break;
        }
        default: break;
        }
        q[0] = bits(kind);
    }
    // The material kind of every sphere, rect and cylinder sits in its cold record too: the shading then knows after ONE
    // dependent load (the primitive's cold record) whether the path ends, scatters or needs a rejection sample, instead
    // of two (cold record -> material record).
    auto kind_of = [&](int material) { return I[(size_t)(L.off_mat + 3 * material) * 4]; };  // the bits, as a float
    for (int k = 0; k < ns_slots; ++k)
        if (slots[k] >= 0) rec4(L.off_sph_cold + k)[3] = kind_of(s.prims[slots[k]].material);
    for (int k = 0; k < L.nr; ++k) rec4(L.off_rect_cold + k)[2] = kind_of(s.prims[rec[k]].material);
    for (int k = 0; k < L.nc; ++k) rec4(L.off_cyl_cold + 4 * k)[14] = kind_of(s.prims[cyl[k]].material);
    c.packed_version = s.version;
    return true;
}

static void pack_scene(const Scene &s, DeviceSceneCache &c) {
    // a cell's list that overflows even the wide tables (more than a thousand primitives through one cell: a clump) moves its
    // members to the always-tested set and the tables are rebuilt: in the limit the scene is scanned, which is the reference's
    // algorithm.  Every round removes at least one primitive from the lists, and real scenes need none.
    std::vector<char> forced(s.prims.size(), 0);
    for (size_t round = 0; round <= s.prims.size(); ++round)
        if (pack_scene_once(s, c, forced)) return;
}

// ---- shard geometry ----------------------------------------------------------------
struct Shard {
    int tile_rows, tile_first, tile_stride, tile_rotate, num_tiles, local_tiles, local_rows;
};

// the k-th row tile of a shard (rt_opts.tile_rotate: plain interleave, rotated interleave, or there-and-back); grows with k
static inline long long shard_tile(const Shard &sh, int k) {
    if (sh.tile_rotate == 2)  // ranks 0 .. N-1, then N-1 .. 0: two tiles per group of 2 N
        return (long long)(k >> 1) * 2 * sh.tile_stride + ((k & 1) ? 2 * sh.tile_stride - 1 - sh.tile_first : sh.tile_first);
    if (!sh.tile_rotate) return sh.tile_first + (long long)k * sh.tile_stride;
    int j = (sh.tile_first - k) % sh.tile_stride;
    if (j < 0) j += sh.tile_stride;
    return (long long)k * sh.tile_stride + j;
}

static int shard_of(const Scene &s, const rt_opts *o, Shard &sh) {
    sh.tile_rows = (o && o->tile_rows > 0) ? o->tile_rows : 8;
    sh.tile_first = o ? o->tile_first : 0;
    sh.tile_stride = (o && o->tile_stride > 1) ? o->tile_stride : 1;
    sh.tile_rotate = (o && sh.tile_stride > 1) ? o->tile_rotate : 0;
    if (sh.tile_rotate < 0 || sh.tile_rotate > 2) {
        set_error("tile_rotate %d: 0 (plain interleave), 1 (rotated) or 2 (there and back)", sh.tile_rotate);
        return RT_ERR_ARG;
    }
    sh.num_tiles = (s.height + sh.tile_rows - 1) / sh.tile_rows;
    if (sh.tile_first < 0 || (sh.tile_stride > 1 && sh.tile_first >= sh.tile_stride)) {
        set_error("tile_first %d out of range for tile_stride %d", sh.tile_first, sh.tile_stride);
        return RT_ERR_ARG;
    }
    sh.local_tiles = 0;
    sh.local_rows = 0;
    // (only the last group of tile_stride tiles can be incomplete, so a shard's tiles are its local tiles 0 .. local_tiles - 1)
    for (int k = 0;; ++k) {
        const long long t64 = shard_tile(sh, k);
        if (t64 >= sh.num_tiles) break;
        const int t = (int)t64;
        int rows = s.height - t * sh.tile_rows;
        if (rows > sh.tile_rows) rows = sh.tile_rows;
        sh.local_rows += rows;
        sh.local_tiles++;
    }
    return RT_OK;
}

static int shard_global_row(const Shard &sh, int local_row) {
    int tl = local_row / sh.tile_rows;
    return (int)shard_tile(sh, tl) * sh.tile_rows + (local_row - tl * sh.tile_rows);
}

}  // namespace rtmi

using namespace rtmi;

extern "C" {

int rt_shard_rows(const rt_scene *s, const rt_opts *o) {
    if (!s) return -RT_ERR_ARG;
    Shard sh;
    int rc = shard_of(s->s, o, sh);
    return rc ? -rc : sh.local_rows;
}

int rt_shard_deal(const rt_scene *s, const rt_opts *o, int n_ranks) {
    if (!s || n_ranks < 1) return -RT_ERR_ARG;
    if (n_ranks == 1) return 0;
    const int tile_rows = (o && o->tile_rows > 0) ? o->tile_rows : 8;
    const long long tiles = (s->s.height + tile_rows - 1) / tile_rows;
    // The rotated interleave evens the ranks out once its phase has gone round a few times (a revolution is n_ranks groups of
    // n_ranks tiles); a frame too short for that is dealt there and back, which cancels the cost's trend inside every 2 n_ranks
    // tiles.  Measured on per-tile query counts of RTIOW 1080p, 135 tiles (tools/gpu_tilecost.py, tile_deal.py), busiest rank
    // above the mean: 4 ranks 0.36 % rotated / 0.86 % there and back, 8 ranks 2.17 % / 0.92 % (plain interleave: 1.16 %, 1.84 %).
    return tiles >= 4LL * n_ranks * n_ranks ? 1 : 2;
}

int rt_shard_global_row(const rt_scene *s, const rt_opts *o, int local_row) {
    if (!s) return -RT_ERR_ARG;
    Shard sh;
    int rc = shard_of(s->s, o, sh);
    if (rc) return -rc;
    if (local_row < 0 || local_row >= sh.local_rows) return -RT_ERR_ARG;
    return shard_global_row(sh, local_row);
}

int rt_shard_scatter_rows(const rt_scene *s, const rt_opts *o, const float *local_rgb, float *full_rgb) {
    if (!s || !local_rgb || !full_rgb) {
        set_error("rt_shard_scatter_rows: null argument");
        return RT_ERR_ARG;
    }
    Shard sh;
    int rc = shard_of(s->s, o, sh);
    if (rc) return rc;
    const size_t row_floats = (size_t)s->s.width * 3;
    for (int lr = 0; lr < sh.local_rows; ++lr)
        memcpy(full_rgb + (size_t)shard_global_row(sh, lr) * row_floats, local_rgb + (size_t)lr * row_floats,
               row_floats * sizeof(float));
    return RT_OK;
}

int rt_has_ablations(void) { return has_ablations() ? 1 : 0; }

int rt_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
        return -RT_ERR_HIP;
    }
    return n;
}

// what rt_opts.variant = 0 stands for in a scene with these tables
static unsigned pick_variant(const RenderParams &P, bool counting, size_t lds_table_bytes) {
    if (!P.grid_wide) return P.grid_sheet ? 2u : 6u;
    const size_t grid_bytes = (size_t)P.hot_vec4_grid * 16, scan_bytes = (size_t)(P.hot_vec4 - (P.off_box - P.off_grid)) * 16;
    // a handful of primitives of several types, none of them listed in a grid: the plain scan is the same search without
    // the per-query set-up (sample_scene.json: 54 against 79 ms)
    if (P.grid_cells == 0 && P.ncl == 0 && !counting && scan_bytes <= lds_table_bytes) return 16u;
    return grid_bytes <= lds_table_bytes ? 36u : 44u;
}

int rt_scene_table_info(const rt_scene *sc, rt_table_info *out) {
    if (!sc || !out) {
        set_error("rt_scene_table_info: null argument");
        return RT_ERR_ARG;
    }
    int rc = scene_validate(sc->s);
    if (rc) return rc;
    Scene &ms = const_cast<Scene &>(sc->s);
    if (!ms.dev) ms.dev = std::make_shared<DeviceSceneCache>();
    DeviceSceneCache &cache = *ms.dev;
    std::lock_guard<std::mutex> lock(cache.mu);
    if (cache.packed_version != sc->s.version) pack_scene(sc->s, cache);
    const RenderParams &L = cache.layout;
    memset(out, 0, sizeof *out);
    out->image_floats = (int32_t)cache.image.size();
    out->grid_wide = L.grid_wide, out->grid_sheet = L.grid_sheet, out->grid_cells = L.grid_cells;
    const float *g = cache.image.data() + (size_t)L.off_grid * 4;
    for (int a = 0; a < 3; ++a) {
        out->grid_min[a] = g[a], out->grid_size[a] = g[8 + a];
        memcpy(&out->grid_n[a], g + 12 + a, 4);
    }
    out->ob_near2 = g[3], out->ob_far2 = g[7];
    out->ns = L.ns, out->np = L.np, out->ncl = L.ncl;
    out->nr = L.nr, out->nc = L.nc, out->nt = L.nt, out->nr_a = L.nr_a, out->nc_a = L.nc_a, out->nt_a = L.nt_a;
    out->off_grid_cells = L.off_grid_cells, out->off_grid_items = L.off_grid_items;
    out->off_sph_cold = L.off_sph_cold, out->off_rect_cold = L.off_rect_cold, out->off_cyl_cold = L.off_cyl_cold, out->off_tri_cold = L.off_tri_cold;
    out->off_rect_hot = L.off_rect_hot, out->off_cyl_hot = L.off_cyl_hot, out->off_tri_hot = L.off_tri_hot;
    out->hot_bytes_grid = L.hot_vec4_grid * 16;
    out->kernel_variant = (int32_t)pick_variant(L, false, (size_t)knob("RTMI_GLOBAL_TABLE_BYTES", (double)kLdsTableBytes));
    return RT_OK;
}

int rt_scene_table_image(const rt_scene *sc, float *dst, int cap_floats) {
    rt_table_info info;
    int rc = rt_scene_table_info(sc, &info);
    if (rc) return -rc;
    if (dst && cap_floats > 0) {
        DeviceSceneCache &cache = *const_cast<Scene &>(sc->s).dev;
        std::lock_guard<std::mutex> lock(cache.mu);
        memcpy(dst, cache.image.data(), sizeof(float) * (size_t)std::min(cap_floats, info.image_floats));
    }
    return info.image_floats;
}

static int render_impl(const rt_scene *sc, const rt_opts *o, void *d_rgb_sum, void *stream_v, rt_stats *stats,
                       long long *h_acc, bool count) {
    if (!sc || (!d_rgb_sum && !h_acc)) {
        set_error("rt_render_hip_device: null scene or output pointer");
        return RT_ERR_ARG;
    }
    const Scene &s = sc->s;
    int rc = scene_validate(s);
    if (rc) return rc;
    Shard sh;
    rc = shard_of(s, o, sh);
    if (rc) return rc;
    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->local_rows = sh.local_rows;
    }
    if (sh.local_rows == 0) return RT_OK;

    int sample_first = o ? o->sample_first : 0;
    int sample_count = (o && o->sample_count > 0) ? o->sample_count : s.spp;
    // samples per work item: scheduling only (the pixel sum is exact).  64 and 128 measure the same on MI355X for
    // frames that fill the chip (199.8 / 200.2 ms; 256: 204.3) and 128 halves the accumulator traffic -- one 1.5 KB
    // tile flush and a handful of orphaned paths per item; small frames get smaller chunks so that there are a few items per
    // resident wave (a 400x225 frame has 1450 tiles for ~6000 resident waves)
    int spp_chunk = (o && o->spp_chunk > 0) ? o->spp_chunk : 0;
    if (spp_chunk == 0) {
        const long long tiles = (long long)((s.width + 7) / 8) * ((sh.local_rows + 7) / 8);
        // 8 items per resident wave, down to 4 samples each: the cost of a tile varies by two orders of magnitude (sky against a
        // triangle mesh), and a wave's share evens out only over several items (20 000 triangles at 1280 x 720 x 16: 31.2 ms
        // with 4 items of 8 samples per wave, 24.6 with 8 of 4; 20 000 spheres 10.1 either way)
        const long long want_items = 8LL * 256 * RT_WAVES_PER_SIMD * 4;
        // (256 for launches that have the items: with the graded tail below the whole 1080p x 1024 spp frame takes the same 119.3 -
        //  119.7 ms with 128- or 256-sample items (512: 120.0); the tail's medium and small items keep the item count -- 16 per tile
        //  against 14 -- and with it the accumulator flushes where they were: 1.28 GB of HBM writes per frame, PMC)
        spp_chunk = 256;
        while (spp_chunk > 4 && tiles * ((sample_count + spp_chunk - 1) / spp_chunk) < want_items) spp_chunk /= 2;
    }
    if (spp_chunk > sample_count) spp_chunk = sample_count;
    // Guided self-scheduling of the chunk-major queue: big chunks first, then chunks a quarter as long, then
    // a sixteenth.  A launch ends when the last wave finishes its last item; a 64-sample item takes ~3 ms of
    // wall time (seven waves share a SIMD) and an item over glass and dense spheres several times the average,
    // so the runs of shorter items have to last long enough for the other waves to have something to do
    // meanwhile.  How many big chunks are given up follows from r = resident waves / tiles: a whole 1080p
    // frame (r = 0.22) gives up one of four 256-sample chunks, a 1/8 row shard (r = 1.8) twelve of sixteen 64-sample ones; of
    // the medium chunks that many again are cut into small ones.  Short items cost little since stragglers no longer
    // block a wave's next item (chunk sizes 16..128 measure within 1.5 % on the whole frame).
    // (knob(): measurement knobs of the default build, constants in a product build)
    static const int tail_mode = (int)knob("RTMI_TAIL_MODE", 1);  // 0 = one run of equal chunks
    // (round 2: 4 / 6 / 8 / 12 -> a 1/4 shard 66.0 / 63.8 / 63.2 / 61.8 ms.  Round 3, four times as many items, RTIOW 1080p x 1024 spp
    //  (tools/gpu_tail_sweep.py): 5 / 6 / 7 / 8 / 12 -> rank 0's 1/8 shard 16.70 / 16.32 / 16.34 / 16.41 / 16.53 ms, a 1/4 shard 31.62 /
    //  31.21 / 31.42 / 31.49 / 31.63, the whole frame within 0.2 %; at 4 and below the last big items outlast the short ones: 17.7 ms)
    //  Round 3's last session (tools/gpu_tail_sweep.py; whole frame / a 1/2 / a 1/4 / rank 0's and rank 3's 1/8 shard, ms): with a run of
    //  small items for the big launches too (below) 120.35 / 61.60 / 31.06 / 16.13 / 15.49 -> 119.52 / 60.45 / 31.05 / 16.15 / 15.51.
    //  The number of big chunks given up rounded DOWN as well (one for a whole 1080p frame instead of two; 3 / 6 / 12 for the shards): 119.72 /
    //  60.98 / 31.09 / 16.15 / 15.54 -> 119.42 / 60.72 / 30.97 / 16.15 / 15.48, the DNA frame 6.61 -> 6.52, 20 000 triangles 26.8 -> 25.5 ms;
    //  the cliff (factor 5 rounded down: four chunks for a 1/4 shard, 32.06 ms) is two chunks away.)
    static const double tail_factor = knob("RTMI_TAIL_FACTOR", 7.0);
    static const int tail_div = std::max(2, (int)knob("RTMI_TAIL_DIV", 4));  // big : medium item length
    static const int orphan_env = (int)knob("RTMI_ORPHAN_MAX", -1);
    int n_big = sample_count / spp_chunk, n_med = 0, q_med = spp_chunk, q_small = spp_chunk;
    int num_chunks, orphan_max = 12;
    {
        const long long tiles = (long long)((s.width + 7) / 8) * ((sh.local_rows + 7) / 8);
        const double r = 256.0 * 4 * RT_WAVES_PER_SIMD / (double)std::max(1LL, tiles);
        const int rem = sample_count - n_big * spp_chunk;
        int rest = rem;  // samples after the big chunks
        if (tail_mode > 0 && spp_chunk >= 16 && n_big >= 1) {
            static const int tail_floor = (int)knob("RTMI_TAIL_FLOOR", 1);  // (0: round up, as until the end of round 3)
            const int n_split = std::min(n_big, std::max(1, tail_floor ? (int)std::floor(tail_factor * r) : (int)std::ceil(tail_factor * r)));  // big chunks given up
            n_big -= n_split;
            rest += n_split * spp_chunk;
            q_med = std::max(4, spp_chunk / tail_div);
            q_small = std::max(4, q_med / 4);
            int small_samples = 0;
            // a run of small items behind the medium ones: for launches of few tiles per wave, and (round 3's end) for every launch whose
            // small items are still 16 samples long (the whole 1080p x 1024 spp frame 120.35 -> 119.5 ms, a 1/2 shard 61.6 -> 60.5; with
            // 4-sample items at r = 0.5 the DNA frame goes from 6.6 to 7.7 ms: 400 000 items for 6 ms of work)
            static const double small_r = knob("RTMI_SMALL_R", 0.5);
            if ((r >= small_r || q_small >= 16) && q_small < q_med)
                small_samples = std::min(rest - q_med, std::max(1, (int)std::ceil(tail_factor * r)) * q_med);
            if (small_samples < 0) small_samples = 0;
            n_med = (rest - small_samples) / q_med;  // whole medium chunks; the small run takes what is left
        } else {
            q_med = q_small = spp_chunk;  // one run of equal chunks (the remainder is the last, shorter one)
        }
        const int after_med = rest - n_med * q_med;
        num_chunks = n_big + n_med + (after_med + q_small - 1) / q_small;
        // orphans: waiting for an item's last paths costs short items more (render_kernel.hip, step 4)
        orphan_max = orphan_env >= 0 ? std::min(63, orphan_env) : (r >= 0.5 ? 63 : 12);
    }
    if (sample_first < 0) {
        set_error("sample_first must be >= 0");
        return RT_ERR_ARG;
    }
    // a sample contributes at most 2^16 (radiance_to_fixed clamps) in units of 2^-24: 2^23 samples keep the 64-bit
    // pixel sums exact (the reference's float sum saturates gracefully instead; a silent wrap here would not)
    if ((long long)sample_first + sample_count > RT_MAX_SAMPLES_PER_PIXEL) {
        set_error("samples [%d, %lld) exceed %d samples per pixel, the range over which the fixed-point pixel sums are exact",
                  sample_first, (long long)sample_first + sample_count, RT_MAX_SAMPLES_PER_PIXEL);
        return RT_ERR_LIMIT;
    }

    unsigned variant = o ? o->variant : 0;
    if (!variant_exists(variant)) {
        set_error("unknown kernel variant %u%s", variant, has_ablations() ? "" : " (this library was built without the measurement variants: make ABLATIONS=1)");
        return RT_ERR_ARG;
    }
    if (count && !has_ablations()) {
        set_error("rt_render_hip_count: this library was built without the counting kernels (make ABLATIONS=1)");
        return RT_ERR_LIMIT;
    }
    // triangles and image textures live in separate builds of the kernels (template argument EXT)
    bool ext = !s.images.empty();
    for (const rt_prim &p : s.prims) ext = ext || p.type == RT_PRIM_TRIANGLE;
    const bool force_ext = knob_set("RTMI_FORCE_EXT");  // measurement: the EXT builds on scenes that do not need them
    int device = o ? o->device : 0;
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (ndev <= 0) {
        set_error("no HIP device visible: the render path has no CPU fallback");
        return RT_ERR_HIP;
    }
    if (device < 0 || device >= ndev) {
        set_error("device %d out of range (%d visible)", device, ndev);
        return RT_ERR_ARG;
    }
    int prev_device = 0;
    HIP_TRY(hipGetDevice(&prev_device));
    if (prev_device != device) HIP_TRY(hipSetDevice(device));
    struct Restore {
        int prev, cur;
        ~Restore() {
            if (prev != cur) (void)hipSetDevice(prev);
        }
    } restore{prev_device, device};

    hipStream_t stream = (hipStream_t)stream_v;

    // ---- resident scene image
    Scene &ms = const_cast<Scene &>(s);
    if (!ms.dev) ms.dev = std::make_shared<DeviceSceneCache>();
    DeviceSceneCache &cache = *ms.dev;
    // The lock covers packing, the entry list and the enqueueing of this call's work; it is released before the
    // call waits for the device (stats), so that threads which render one scene on DIFFERENT devices overlap.
    std::unique_lock<std::mutex> lock(cache.mu);
    if (cache.packed_version != s.version) pack_scene(s, cache);
    DeviceEntry *ent = nullptr;
    for (auto &e : cache.entries)
        if (e->device == device) ent = e.get();
    if (!ent) {
        cache.entries.emplace_back(new DeviceEntry());
        ent = cache.entries.back().get();
        ent->device = device;
    }
    // timing events: three per (host thread, device), created at the thread's first timed call there and kept
    struct Events {
        std::vector<std::array<hipEvent_t, 3>> per_device;
        ~Events() {
            for (auto &t : per_device)
                for (hipEvent_t ev : t)
                    if (ev) (void)hipEventDestroy(ev);
        }
    };
    static thread_local Events events;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;
    if (stats) {
        if ((int)events.per_device.size() <= device) events.per_device.resize((size_t)device + 1, {nullptr, nullptr, nullptr});
        auto &t = events.per_device[(size_t)device];
        for (hipEvent_t &ev : t)
            if (!ev) HIP_TRY(hipEventCreate(&ev));
        ev0 = t[0], ev1 = t[1], ev2 = t[2];
        HIP_TRY(hipEventRecord(ev0, stream));
    }
    size_t image_bytes = cache.image.size() * sizeof(float);
    if (ent->version != s.version || !ent->d_image) {
        if (ent->image_bytes < image_bytes) {
            if (ent->d_image) HIP_TRY(hipFree(ent->d_image));
            ent->d_image = nullptr;
            HIP_TRY(hipMalloc(&ent->d_image, image_bytes));
            ent->image_bytes = image_bytes;
        }
        // stream-ordered: the launches below follow on the same stream (renders of one scene object on one device share
        // a stream: include/rtmi.h).  The source is pageable, so the call returns once the bytes are staged.
        HIP_TRY(hipMemcpyAsync(ent->d_image, cache.image.data(), image_bytes, hipMemcpyHostToDevice, stream));
        ent->version = s.version;
    }

    // ---- kernel parameters
    RenderParams P = cache.layout;
    for (int i = 0; i < 3; ++i) P.background[i] = s.background[i];
    P.flags = s.flags;
    P.rr_p = s.rr_p;
    P.width = s.width, P.height = s.height, P.max_depth = s.max_depth;
    P.inv_wm1 = 1.0f / (float)(s.width - 1), P.inv_hm1 = 1.0f / (float)(s.height - 1);  // IEEE fp32 divisions, as the checker's
    P.tile_rows = sh.tile_rows, P.tile_first = sh.tile_first, P.tile_stride = sh.tile_stride;
    P.num_tiles = sh.num_tiles, P.local_rows = sh.local_rows;
    P.sample_first = sample_first, P.sample_count = sample_count;
    P.spp_chunk = spp_chunk, P.num_chunks = num_chunks;
    P.n_big = n_big, P.n_med = n_med, P.q_med = q_med, P.q_small = q_small;
    P.orphan_max = orphan_max;
    uint64_t seed = o ? o->seed : 0;
    P.seed_lo = (uint32_t)seed, P.seed_hi = (uint32_t)(seed >> 32);
    P.tiles_x = (s.width + 7) / 8;
    P.bands = (sh.local_rows + 7) / 8;

    // ---- which kernel.  LDS per workgroup: the hot tables (unless the variant reads them from global memory: bit 3) + one
    // tile accumulator per wave.  The tables live in LDS while that leaves room for the kernel's full occupancy
    // (RT_WAVES_PER_SIMD workgroups per CU); larger scenes run the same walk over global memory (4000 spheres 6.7 vs 19 ms),
    // which has no size limit.  The packer has chosen the table format (device_scene.h): COMPACT for sphere-only scenes whose
    // tables fit that LDS budget, WIDE for every other scene.
    const size_t acc_lds = kAccLds;
    auto hot_bytes_of = [&](unsigned v) {  // (each candidate search stages the part of the hot tables it reads)
        const int mode = variant_cull_mode(v);
        if (mode == 5 || mode == 6 || mode == 7) return (size_t)P.hot_vec4_grid * 16;
        return (size_t)((mode == 3 ? P.hot_vec4_tables : P.hot_vec4) - (P.off_box - P.off_grid)) * 16;  // (without the grid tables)
    };
    static const size_t global_threshold = (size_t)knob("RTMI_GLOBAL_TABLE_BYTES", (double)kLdsTableBytes);
    const bool sphere_only = P.nr + P.nc + P.nt == 0 && !ext;
    auto pick = [&](bool counting) -> unsigned { return pick_variant(P, counting, global_threshold); };
    if (variant == 0) variant = pick(count);
    // the counting kernels exist for the grid walks (3-D) and two ablation searches: anything else is counted by the kernel
    // variant 0 would run (reported in stats->kernel_variant / cull_mode)
    if (count && !variant_has_count(variant)) variant = (variant == 2) ? 6u : pick(true);
    if (count && variant == 2) variant = 6;
    const int mode = variant_cull_mode(variant);
    if ((mode == 5 || mode == 6) && P.grid_wide) {
        set_error("kernel variant %u reads the compact grid tables of a sphere-only scene that fits LDS; this scene has the wide ones "
                  "(other primitives, textures, 65536 sphere slots or more, or tables beyond %zu bytes): use variant 0, 36 or 44",
                  variant, global_threshold);
        return RT_ERR_LIMIT;
    }
    if (mode == 7 && !P.grid_wide) {
        set_error("kernel variant %u walks the wide grid tables; this scene (spheres only, small enough for LDS) has the compact ones: "
                  "use variant 0, 2 or 6", variant);
        return RT_ERR_LIMIT;
    }
    if (variant == 2 && !P.grid_sheet) {
        set_error("kernel variant 2 walks a grid that is one cell high, which this scene does not have");
        return RT_ERR_LIMIT;
    }
    if (ext && !variant_has_ext(variant)) {
        set_error("kernel variant %u has no build with triangles / image textures (variants 0, 36, 44 and the linear scans 16 / 24 have)", variant);
        return RT_ERR_LIMIT;
    }
    if (!sphere_only && (mode == 5 || mode == 6)) {  // (cannot happen: such scenes get wide tables)
        set_error("kernel variant %u is built for sphere-only scenes", variant);
        return RT_ERR_LIMIT;
    }
    if (force_ext && variant_has_ext(variant) && P.grid_wide) ext = true;
    const size_t hot_bytes = hot_bytes_of(variant);
    const size_t lds_bytes = ((variant & 8u) ? 0 : hot_bytes) + acc_lds;
    if (knob_set("RTMI_DEBUG_LAYOUT")) {
        const float *g = cache.image.data() + (size_t)P.off_grid * 4;
        int gn[3];
        memcpy(gn, g + 12, sizeof gn);
        fprintf(stderr, "variant %u: LDS %zu bytes per workgroup (tables %zu); %d prefix slots, %d clusters of %d; always-tested others %d + %d + %d "
                "of %d + %d + %d; %s grid %d x %d x %d = %d cells, cell %.3f x %.3f x %.3f (vec4 records: cells %d, lists %d)\n",
                variant, lds_bytes, (variant & 8u) ? (size_t)0 : hot_bytes, P.np, P.ncl, P.cluster, P.nr_a, P.nc_a, P.nt_a, P.nr, P.nc, P.nt,
                P.grid_wide ? "wide" : "compact", gn[0], gn[1], gn[2], P.grid_cells,
                g[8], g[9], g[10], P.off_grid_items - P.off_grid_cells, P.hot_vec4_grid - P.off_grid_items);
    }
    if (lds_bytes > 160 * 1024) {
        set_error("kernel variant %u keeps the scene tables in LDS and this scene needs %zu bytes per workgroup "
                  "(limit 163840); use the default variant",
                  variant, lds_bytes);
        return RT_ERR_LIMIT;
    }
    if (lds_bytes > 64 * 1024 && set_max_dynamic_lds(lds_bytes)) {
        set_error("cannot raise the dynamic LDS limit to %zu bytes", lds_bytes);
        return RT_ERR_HIP;
    }
    if (s.width > 65536 || P.bands > 32767) {  // the kernel packs (tile x0, band) of a wave's older item into one register
        set_error("frame of %d x %d rows per shard exceeds the tile index range (65536 columns, 262136 rows)", s.width,
                  sh.local_rows);
        return RT_ERR_LIMIT;
    }
    const size_t plane = (size_t)sh.local_rows * s.width * 3;
    const unsigned long long items64 = (unsigned long long)P.tiles_x * P.bands * num_chunks;
    if (items64 > 0x7fffffffull) {
        set_error("%llu work items exceed the queue counter", items64);
        return RT_ERR_LIMIT;
    }
    P.num_items = (int)items64;
    // persistent launch: enough workgroups to fill the chip, never more than the work needs
    if (ent->num_cus == 0) {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device));
        ent->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const unsigned long long resident = (unsigned long long)ent->num_cus * blocks_per_cu(variant, count, lds_bytes, ext);
    const unsigned long long need_blocks = (items64 + 3) / 4;
    const unsigned long long grid64 = need_blocks < resident ? (need_blocks ? need_blocks : 1) : resident;

    float *d_out = (float *)d_rgb_sum;
    DevCounters *d_cnt = nullptr;
    if (count) {
        if (!ent->d_counters) HIP_TRY(hipMalloc((void **)&ent->d_counters, sizeof(DevCounters)));
        d_cnt = ent->d_counters;
        HIP_TRY(hipMemsetAsync(d_cnt, 0, sizeof(DevCounters), stream));
        // the two minima start at all-ones
        HIP_TRY(hipMemsetAsync(&d_cnt->t_start_min, 0xff, sizeof(unsigned long long), stream));
        HIP_TRY(hipMemsetAsync(&d_cnt->t_end_min, 0xff, sizeof(unsigned long long), stream));
        HIP_TRY(hipMemsetAsync(&d_cnt->t_qe_min, 0xff, sizeof(unsigned long long), stream));
    }
    if (stats) HIP_TRY(hipEventRecord(ev1, stream));

    int launches = 0;
    if (s.max_depth <= 0 && !h_acc) {
        // while (depth > 0) never runs: every sample is black (main.cpp:20,42)
        HIP_TRY(hipMemsetAsync(d_out, 0, plane * sizeof(float), stream));
    } else {
        // accumulators + the work-queue counter behind them, cleared together.  They belong to this
        // (scene, device): concurrent renders of ONE scene object on one device must share a stream
        // (different scene objects, or clones, are independent)
        // accumulators, then (256-byte aligned: the kernel reads the ItemParams with 16-byte loads) the queue
        // counter and the ItemParams
        const size_t queue_off = (plane * sizeof(unsigned long long) + 255) & ~(size_t)255;
        const size_t need = queue_off + 256;
        if (ent->acc_bytes < need) {
            if (ent->d_acc) HIP_TRY(hipFree(ent->d_acc));
            ent->d_acc = nullptr;
            ent->acc_bytes = 0;
            HIP_TRY(hipMalloc((void **)&ent->d_acc, need));
            ent->acc_bytes = need;
        }
        HIP_TRY(hipMemsetAsync(ent->d_acc, 0, need, stream));
        // progressive rendering: continue from the caller's exact sums
        if (h_acc) HIP_TRY(hipMemcpyAsync(ent->d_acc, h_acc, plane * sizeof(long long), hipMemcpyHostToDevice, stream));
        unsigned int *d_queue = reinterpret_cast<unsigned int *>(reinterpret_cast<char *>(ent->d_acc) + queue_off);
        {  // what a wave reads when it fetches or flushes an item (kept out of the kernel's SGPRs)
            ItemParams ip;
            ip.tiles_x = P.tiles_x, ip.bands = P.bands, ip.num_items = P.num_items;
            ip.sample_first = P.sample_first, ip.sample_count = P.sample_count, ip.spp_chunk = P.spp_chunk;
            ip.n_big = P.n_big, ip.n_med = P.n_med, ip.q_med = P.q_med, ip.q_small = P.q_small;
            ip.tile_rows = P.tile_rows, ip.tile_first = P.tile_first, ip.tile_stride = P.tile_stride;
            ip.tile_rotate = sh.tile_rotate;
            ip.local_rows = P.local_rows;
            launch_item_params(d_queue, ip, stream);
        }
        if (s.max_depth > 0) {
            if (!launch_render(P, ent->d_image, ent->d_acc, d_queue, d_cnt, lds_bytes, (unsigned)grid64, stream, variant, ext)) {
                set_error("kernel variant %u has no %s build", variant, count ? "counting" : (ext ? "triangle / texture" : "such"));
                return RT_ERR_LIMIT;
            }
            ++launches;
        }
        if (d_out) {
            launch_finalize(ent->d_acc, d_out, plane, stream);
            ++launches;
        }
        if (h_acc) {
            HIP_TRY(hipMemcpyAsync(h_acc, ent->d_acc, plane * sizeof(long long), hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
        }
    }
    HIP_TRY(hipGetLastError());

    if (stats) {
        stats->kernel_variant = (int32_t)variant;  // what variant 0 (or a counting call) resolved to
        HIP_TRY(hipEventRecord(ev2, stream));
        lock.unlock();
        HIP_TRY(hipEventSynchronize(ev2));
        float up = 0, k = 0;
        HIP_TRY(hipEventElapsedTime(&up, ev0, ev1));
        HIP_TRY(hipEventElapsedTime(&k, ev1, ev2));
        stats->upload_ms = up;
        stats->kernel_ms = k;
        stats->launches = launches;
        if (count) {
            DevCounters h;
            HIP_TRY(hipMemcpy(&h, d_cnt, sizeof h, hipMemcpyDeviceToHost));
            stats->samples = h.samples;
            stats->queries = h.queries;
            stats->prim_tests = h.queries * (unsigned long long)s.prims.size();
            stats->hits = h.hits;
            stats->misses = h.misses;
            for (int i = 0; i < 4; ++i) stats->scatter[i] = h.scatter[i];
            stats->rng_draws = h.rng_draws;
            stats->cand_lanes = h.cand_lanes;
            stats->cand_waves = h.cand_waves;
            stats->clusters_visited = h.clusters_visited;
            stats->groups_visited = h.groups_visited;
            stats->lane_clusters = h.lane_clusters;
            stats->lane_groups = h.lane_groups;
            stats->lane_cands = h.lane_cands;
            stats->group_maxpop = h.group_maxpop;
            stats->query_maxpop = h.query_maxpop;
            for (int i = 0; i < 6; ++i) stats->cycles[i] = h.cycles[i];
            stats->wave_start_spread_us = (double)(h.t_start_max - h.t_start_min) * 0.01;
            stats->wave_end_spread_us = (double)(h.t_end_max - h.t_end_min) * 0.01;
            stats->wave_span_us = (double)(h.t_end_max - h.t_start_min) * 0.01;
            if (knob_set("RTMI_DEBUG_DRAIN")) {
                fprintf(stderr, "shader clock over the waves' lifetimes: %.0f MHz\n", h.life_ticks ? 100.0 * (double)h.life_cycles / (double)h.life_ticks : 0.0);
                fprintf(stderr, "queue-empty seen over %.1f us; first exit %.1f us after the first queue-empty; drain histogram (50 us bins):",
                        (double)(h.t_qe_max - h.t_qe_min) * 0.01, (double)(h.t_end_min - h.t_qe_min) * 0.01);
                for (int i = 0; i < 32; ++i) fprintf(stderr, " %u", h.drain_hist[i]);
                for (int e = 0; e < 2; ++e) {
                    fprintf(stderr, "\nwave-queries by live lanes (bins of 4 lanes, last = all 64), %s:", e ? "after the wave found the queue empty" : "while the queue had items");
                    for (int i = 0; i < 17; ++i) fprintf(stderr, " %llu", h.occ_hist[e][i]);
                }
                fprintf(stderr, "\nwaves by time from start to queue-empty (64 us bins, first nonzero bin on):");
                int first = 0;
                while (first < 1023 && !h.qe_hist[first]) ++first;
                fprintf(stderr, " [bin %d]", first);
                for (int i = first; i < 1024 && i < first + 60; ++i) fprintf(stderr, " %u", h.qe_hist[i]);
                fprintf(stderr, "\nwaves by time from start to exit (same bins):");
                for (int i = first; i < 1024 && i < first + 60; ++i) fprintf(stderr, " %u", h.exit_hist[i]);
                fprintf(stderr, "\n");
            }
            stats->wave_queries = h.wave_queries;
            stats->cull_prefix = P.np, stats->cull_clusters = P.ncl, stats->cull_groups = P.ngr;
            stats->cull_cluster_size = P.cluster;
            stats->cull_mode = variant_cull_mode(variant), stats->cull_windows = P.nwin;  // (of the kernel that ran)
            stats->grid_sheet = P.grid_sheet;
        }
    }
    return RT_OK;
}

int rt_render_hip_device(const rt_scene *s, const rt_opts *o, void *d_rgb_sum, void *stream, rt_stats *stats) {
    return render_impl(s, o, d_rgb_sum, stream, stats, nullptr, false);
}

static int render_host_buffer(const rt_scene *sc, const rt_opts *o, float *rgb_sum, rt_stats *stats, bool count,
                              long long *h_acc = nullptr) {
    if (!sc) {
        set_error("null scene");
        return RT_ERR_ARG;
    }
    if (count && !stats) {
        set_error("rt_render_hip_count needs a stats pointer");
        return RT_ERR_ARG;
    }
    if (!count && !rgb_sum && !h_acc) {
        set_error("null output buffer");
        return RT_ERR_ARG;
    }
    Shard sh;
    int rc = shard_of(sc->s, o, sh);
    if (rc) return rc;
    int device = o ? o->device : 0;
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (ndev <= 0) {
        set_error("no HIP device visible: the render path has no CPU fallback");
        return RT_ERR_HIP;
    }
    if (device < 0 || device >= ndev) {
        set_error("device %d out of range (%d visible)", device, ndev);
        return RT_ERR_ARG;
    }
    int prev = 0;
    HIP_TRY(hipGetDevice(&prev));
    if (prev != device) HIP_TRY(hipSetDevice(device));
    struct Restore {  // the caller's device comes back on every return path
        int prev, cur;
        ~Restore() {
            if (prev != cur) (void)hipSetDevice(prev);
        }
    } restore{prev, device};
    const size_t bytes = (size_t)sh.local_rows * sc->s.width * 3 * sizeof(float);
    // the device framebuffer of this (scene, device) is kept between calls (no hipMalloc / hipFree per frame)
    float *d_out = nullptr;
    rt_stats local;
    if (bytes && (rgb_sum || !h_acc)) {
        Scene &ms = const_cast<Scene &>(sc->s);
        if (!ms.dev) ms.dev = std::make_shared<DeviceSceneCache>();
        DeviceSceneCache &cache = *ms.dev;
        std::lock_guard<std::mutex> lock(cache.mu);
        DeviceEntry *ent = nullptr;
        for (auto &e : cache.entries)
            if (e->device == device) ent = e.get();
        if (!ent) {
            cache.entries.emplace_back(new DeviceEntry());
            ent = cache.entries.back().get();
            ent->device = device;
        }
        if (ent->out_bytes < bytes) {
            if (ent->d_out) HIP_TRY(hipFree(ent->d_out));
            ent->d_out = nullptr, ent->out_bytes = 0;
            HIP_TRY(hipMalloc((void **)&ent->d_out, bytes));
            ent->out_bytes = bytes;
        }
        d_out = ent->d_out;
    }
    rc = bytes ? render_impl(sc, o, d_out, nullptr, stats ? stats : &local, h_acc, count) : RT_OK;
    if (rc == RT_OK && rgb_sum && bytes) {
        hipError_t e = hipMemcpy(rgb_sum, d_out, bytes, hipMemcpyDeviceToHost);
        if (e != hipSuccess) {
            set_error("hipMemcpy D2H failed: %s", hipGetErrorString(e));
            rc = RT_ERR_HIP;
        }
    }
    return rc;
}

int rt_render_hip(const rt_scene *s, const rt_opts *o, float *rgb_sum, rt_stats *stats) {
    return render_host_buffer(s, o, rgb_sum, stats, false);
}

int rt_render_hip_count(const rt_scene *s, const rt_opts *o, float *rgb_sum, rt_stats *stats) {
    return render_host_buffer(s, o, rgb_sum, stats, true);
}

int rt_render_hip_accumulate(const rt_scene *s, const rt_opts *o, int64_t *acc, float *rgb_sum, rt_stats *stats) {
    if (!acc) {
        set_error("rt_render_hip_accumulate: null accumulator");
        return RT_ERR_ARG;
    }
    static_assert(sizeof(long long) == sizeof(int64_t), "accumulator width");
    return render_host_buffer(s, o, rgb_sum, stats, false, reinterpret_cast<long long *>(acc));
}

}  // extern "C"
