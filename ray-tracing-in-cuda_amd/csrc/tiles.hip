// One frame on several GPUs of one node behind the C ABI: rt_render_hip_tiles (include/rtmi.h).
//
// The reference's only multi-GPU mechanism is one renderer PROCESS per GPU per animation frame
// (gpu-version/blue.py:23-32, CUDA_VISIBLE_DEVICES=k); it has no collective.  Here one frame is split:
// row tile t (tile_rows full-width rows) belongs to device t mod N, every device renders its tiles into a
// dense local buffer on its own stream (the same rt_render_hip_device a single GPU runs, shard geometry in
// rt_opts), ONE ncclGather (rccl.h:745; xGMI peer-to-peer, one message per peer) brings the buffers to the
// first device, a small kernel there puts the rows where they belong, and one copy hands the frame to the
// caller.  Single process, single host thread: every launch is asynchronous, so the devices run concurrently.
//
// RCCL is looked up at run time (dlopen) the first time more than the render is needed: a process that has
// already loaded an RCCL (PyTorch bundles one) keeps using that one, and single-GPU consumers of librtmi.so
// carry no RCCL dependency.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>

#include "scene.hpp"

namespace rtmi {

// ---- the few RCCL entry points used: types and signatures from <rccl/rccl.h>, addresses from dlsym (so the
// library is NOT a link-time dependency of librtmi.so)
struct RcclApi {
    void *handle = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGather) Gather = nullptr;  // rccl.h:745
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

static RcclApi *rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (api.handle) break;
        }
        if (!api.handle) return;
        auto sym = [&](const char *name) { return dlsym(api.handle, name); };
        api.CommInitAll = (decltype(api.CommInitAll))sym("ncclCommInitAll");
        api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
        api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
        api.Gather = (decltype(api.Gather))sym("ncclGather");
        api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
        if (!api.CommInitAll || !api.CommDestroy || !api.GroupStart || !api.GroupEnd || !api.Gather || !api.GetErrorString) {
            dlclose(api.handle);
            api.handle = nullptr;
        }
    });
    return api.handle ? &api : nullptr;
}

#define HIP_TRY(expr)                                                                                                   \
    do {                                                                                                                \
        hipError_t e_ = (expr);                                                                                         \
        if (e_ != hipSuccess) {                                                                                         \
            set_error("HIP error %d (%s) at %s:%d: %s", (int)e_, hipGetErrorString(e_), __FILE__, __LINE__, #expr);     \
            return RT_ERR_HIP;                                                                                          \
        }                                                                                                               \
    } while (0)
#define NCCL_TRY(api, expr)                                                                                             \
    do {                                                                                                                \
        ncclResult_t r_ = (expr);                                                                                       \
        if (r_ != ncclSuccess) {                                                                                                  \
            set_error("RCCL error %d (%s) at %s:%d: %s", (int)r_, (api)->GetErrorString(r_), __FILE__, __LINE__, #expr); \
            return RT_ERR_HIP;                                                                                          \
        }                                                                                                               \
    } while (0)

// gathered[rank][pad_rows][W][3] (each rank's local rows dense; which rank owns tile t of the frame, and as which of its
// local tiles, follows rt_opts.tile_rotate: 0 plain interleave, 1 rotated, 2 there and back)  ->  full[H][W][3].  One thread
// per float; consecutive threads read and write consecutive floats.
__global__ __launch_bounds__(256) void place_rows_kernel(const float *__restrict__ gathered, float *__restrict__ full,
                                                         int height, int row_floats, int tile_rows, int n_ranks,
                                                         int pad_rows, int rotate) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)height * row_floats;
    if (i >= total) return;
    const int y = (int)(i / row_floats);
    const int c = (int)(i - (size_t)y * row_floats);
    const int t = y / tile_rows;
    int rank = t % n_ranks, local_tile = t / n_ranks;
    if (rotate == 1) {
        rank = (t % n_ranks + t / n_ranks) % n_ranks;
    } else if (rotate == 2) {
        const int p = t % (2 * n_ranks), back = p >= n_ranks ? 1 : 0;
        rank = back ? 2 * n_ranks - 1 - p : p;
        local_tile = 2 * (t / (2 * n_ranks)) + back;
    }
    const int local_row = local_tile * tile_rows + (y - t * tile_rows);
    full[i] = gathered[((size_t)rank * pad_rows + local_row) * row_floats + c];
}

// Per device-list state that outlives a call: streams, communicators (ncclCommInitAll costs ~100 ms), buffers.
struct TileGroup {
    std::vector<int> devices;
    std::vector<hipStream_t> streams;
    std::vector<ncclComm_t> comms;       // empty until the first gather
    std::vector<float *> local;          // per device: pad_rows x W x 3
    std::vector<hipEvent_t> ev0, ev1;    // render span per device
    size_t local_floats = 0;
    float *gathered = nullptr;           // root: N x local_floats
    size_t gathered_floats = 0;
    float *full = nullptr;               // root: H x W x 3
    size_t full_floats = 0;
    hipEvent_t ev_gather0 = nullptr, ev_done = nullptr;  // root stream
    std::mutex mu;                       // one frame at a time per group

    ~TileGroup() { release(); }
    // frees everything and leaves the group as new (also after a failed set-up, so that the next call starts over instead of
    // finding half of the handles)
    void release() {
        int cur = 0;
        const bool have = hipGetDevice(&cur) == hipSuccess;
        RcclApi *api = comms.empty() ? nullptr : rccl_api();
        for (size_t i = 0; i < devices.size(); ++i) {
            if (hipSetDevice(devices[i]) != hipSuccess) continue;
            if (api && i < comms.size() && comms[i]) (void)api->CommDestroy(comms[i]);
            if (i < local.size() && local[i]) (void)hipFree(local[i]);
            if (i < ev0.size() && ev0[i]) (void)hipEventDestroy(ev0[i]);
            if (i < ev1.size() && ev1[i]) (void)hipEventDestroy(ev1[i]);
            if (i == 0) {
                if (gathered) (void)hipFree(gathered);
                if (full) (void)hipFree(full);
                if (ev_gather0) (void)hipEventDestroy(ev_gather0);
                if (ev_done) (void)hipEventDestroy(ev_done);
            }
            if (i < streams.size() && streams[i]) (void)hipStreamDestroy(streams[i]);
        }
        if (have) (void)hipSetDevice(cur);
        streams.clear(), comms.clear(), local.clear(), ev0.clear(), ev1.clear();
        local_floats = gathered_floats = full_floats = 0;
        gathered = full = nullptr;
        ev_gather0 = ev_done = nullptr;
    }
    // waits for whatever has been queued on the group's streams (error paths: nothing of a failed frame is left running)
    void drain() {
        for (size_t i = 0; i < devices.size() && i < streams.size(); ++i)
            if (streams[i] && hipSetDevice(devices[i]) == hipSuccess) (void)hipStreamSynchronize(streams[i]);
    }
};

// never destroyed at process exit (the HIP runtime and RCCL may already be gone by then); rt_tiles_shutdown() frees
static std::mutex g_groups_mu;
// (shared ownership: a frame in flight keeps its group alive across rt_tiles_shutdown)
static std::vector<std::shared_ptr<TileGroup>> &g_groups = *new std::vector<std::shared_ptr<TileGroup>>();

static std::shared_ptr<TileGroup> group_for(const std::vector<int> &devices) {
    std::lock_guard<std::mutex> lock(g_groups_mu);
    for (auto &g : g_groups)
        if (g->devices == devices) return g;
    g_groups.emplace_back(new TileGroup());
    g_groups.back()->devices = devices;
    return g_groups.back();
}

static int ensure(float *&buf, size_t &have, size_t want) {
    if (have >= want) return RT_OK;
    if (buf) HIP_TRY(hipFree(buf));
    buf = nullptr, have = 0;
    HIP_TRY(hipMalloc((void **)&buf, want * sizeof(float)));
    have = want;
    return RT_OK;
}

}  // namespace rtmi

using namespace rtmi;

extern "C" int rt_render_hip_tiles(const rt_scene *sc, const rt_opts *o, const int *devices, int n_devices,
                                   float *rgb_sum, rt_stats *stats) {
    if (!sc || !rgb_sum) {
        set_error("rt_render_hip_tiles: null scene or output buffer");
        return RT_ERR_ARG;
    }
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (ndev <= 0) {
        set_error("no HIP device visible: the render path has no CPU fallback");
        return RT_ERR_HIP;
    }
    if (n_devices < 1 || n_devices > ndev) {
        set_error("rt_render_hip_tiles: %d device(s) requested, %d visible", n_devices, ndev);
        return RT_ERR_ARG;
    }
    std::vector<int> devs(n_devices);
    for (int i = 0; i < n_devices; ++i) {
        devs[i] = devices ? devices[i] : i;
        if (devs[i] < 0 || devs[i] >= ndev) {
            set_error("rt_render_hip_tiles: device %d out of range (%d visible)", devs[i], ndev);
            return RT_ERR_ARG;
        }
        for (int j = 0; j < i; ++j)
            if (devs[j] == devs[i]) {
                set_error("rt_render_hip_tiles: device %d listed twice", devs[i]);
                return RT_ERR_ARG;
            }
    }
    rt_opts base;
    if (o) base = *o;
    else rt_opts_default(&base);
    if (base.tile_rows <= 0) base.tile_rows = 8;
    const int W = sc->s.width, H = sc->s.height;
    const size_t row_floats = (size_t)W * 3;

    int prev = 0;
    HIP_TRY(hipGetDevice(&prev));
    struct Restore {
        int dev;
        ~Restore() { (void)hipSetDevice(dev); }
    } restore{prev};

    const std::shared_ptr<TileGroup> gp = group_for(devs);
    TileGroup *g = gp.get();
    std::lock_guard<std::mutex> frame(g->mu);
    const int N = n_devices;
    // shard geometry: rank r owns row tiles r, r + N, ...; buffers padded to the largest shard (the gather is uniform)
    std::vector<rt_opts> shard(N, base);
    int pad_rows = 0;
    for (int r = 0; r < N; ++r) {
        shard[r].device = devs[r];
        shard[r].tile_first = r;
        shard[r].tile_stride = N;
        shard[r].tile_rotate = rt_shard_deal(sc, &base, N);
        const int rows = rt_shard_rows(sc, &shard[r]);
        if (rows < 0) return -rows;
        pad_rows = std::max(pad_rows, rows);
    }
    const size_t local_floats = (size_t)pad_rows * row_floats;

    // ---- per-device state.  A failure anywhere in the set-up releases the whole group: the next call starts over.
    RcclApi *api = nullptr;
    auto set_up = [&]() -> int {
    if (g->streams.empty()) {
        g->streams.assign(N, nullptr), g->local.assign(N, nullptr), g->ev0.assign(N, nullptr), g->ev1.assign(N, nullptr);
        for (int r = 0; r < N; ++r) {
            HIP_TRY(hipSetDevice(devs[r]));
            HIP_TRY(hipStreamCreateWithFlags(&g->streams[r], hipStreamNonBlocking));
            HIP_TRY(hipEventCreate(&g->ev0[r]));
            HIP_TRY(hipEventCreate(&g->ev1[r]));
        }
        HIP_TRY(hipSetDevice(devs[0]));
        HIP_TRY(hipEventCreate(&g->ev_gather0));
        HIP_TRY(hipEventCreate(&g->ev_done));
    }
    if (g->local_floats < local_floats) {
        for (int r = 0; r < N; ++r) {
            HIP_TRY(hipSetDevice(devs[r]));
            if (g->local[r]) HIP_TRY(hipFree(g->local[r]));
            g->local[r] = nullptr;
            HIP_TRY(hipMalloc((void **)&g->local[r], std::max<size_t>(local_floats, 1) * sizeof(float)));
        }
        g->local_floats = local_floats;
    }
    HIP_TRY(hipSetDevice(devs[0]));
    int rc = ensure(g->gathered, g->gathered_floats, std::max<size_t>((size_t)N * local_floats, 1));
    if (rc) return rc;
    rc = ensure(g->full, g->full_floats, std::max<size_t>((size_t)H * row_floats, 1));
    if (rc) return rc;
    api = rccl_api();
    if (!api) {
        set_error("rt_render_hip_tiles: librccl.so.1 not found (dlopen): the framebuffer gather needs RCCL");
        return RT_ERR_HIP;
    }
    if (g->comms.empty()) {
        std::vector<ncclComm_t> comms(N, nullptr);
        NCCL_TRY(api, api->CommInitAll(comms.data(), N, devs.data()));
        g->comms = comms;
    }
    return RT_OK;
    };
    if (int rc = set_up()) {
        g->drain();
        g->release();
        return rc;
    }

    // ---- the frame.  On an error, what has been queued is waited for before the call returns.
    auto run_frame = [&]() -> int {
    int rc = RT_OK;
    // ---- render: every device its row tiles, asynchronously on its own stream
    for (int r = 0; r < N; ++r) {
        HIP_TRY(hipSetDevice(devs[r]));
        HIP_TRY(hipEventRecord(g->ev0[r], g->streams[r]));
        rc = rt_render_hip_device(sc, &shard[r], g->local[r], (void *)g->streams[r], nullptr);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(g->ev1[r], g->streams[r]));
    }
    // ---- ONE gather to the first device (each rank's call on its own stream, fused in one group)
    HIP_TRY(hipSetDevice(devs[0]));
    HIP_TRY(hipEventRecord(g->ev_gather0, g->streams[0]));
    NCCL_TRY(api, api->GroupStart());
    for (int r = 0; r < N; ++r) {
        ncclResult_t nr = api->Gather(g->local[r], r == 0 ? g->gathered : nullptr, local_floats, ncclFloat, 0, g->comms[r],
                                      g->streams[r]);
        if (nr != ncclSuccess) {
            (void)api->GroupEnd();
            set_error("RCCL error %d (%s) in ncclGather, rank %d", (int)nr, api->GetErrorString(nr), r);
            return RT_ERR_HIP;
        }
    }
    NCCL_TRY(api, api->GroupEnd());
    // ---- rows to their image positions, frame to the caller
    HIP_TRY(hipSetDevice(devs[0]));
    const size_t total = (size_t)H * row_floats;
    if (total) {
        hipLaunchKernelGGL(place_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, g->streams[0], g->gathered,
                           g->full, H, (int)row_floats, base.tile_rows, N, pad_rows, N > 1 ? 1 : 0);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(g->ev_done, g->streams[0]));
    HIP_TRY(hipMemcpyAsync(rgb_sum, g->full, total * sizeof(float), hipMemcpyDeviceToHost, g->streams[0]));
    for (int r = 0; r < N; ++r) {
        HIP_TRY(hipSetDevice(devs[r]));
        HIP_TRY(hipStreamSynchronize(g->streams[r]));
    }
    return RT_OK;
    };
    if (int rc = run_frame()) {
        g->drain();
        return rc;
    }
    if (stats) {
        memset(stats, 0, sizeof *stats);
        float worst = 0.0f;
        for (int r = 0; r < N; ++r) {
            float ms = 0.0f;
            HIP_TRY(hipSetDevice(devs[r]));
            HIP_TRY(hipEventElapsedTime(&ms, g->ev0[r], g->ev1[r]));
            worst = std::max(worst, ms);
        }
        float gms = 0.0f;
        HIP_TRY(hipSetDevice(devs[0]));
        HIP_TRY(hipEventElapsedTime(&gms, g->ev_gather0, g->ev_done));
        stats->kernel_ms = worst;  // the slowest device's render launches
        stats->gather_ms = gms;    // root: from its own render's end to the assembled frame (includes waiting for peers)
        stats->launches = 2 * N + 1;
        stats->local_rows = H;
        stats->devices_used = N;
    }
    return RT_OK;
}

// Device side of rt_shard_scatter_rows for a GATHERED buffer: d_gathered[n_ranks][pad_rows][W][3] (rank r's
// local rows dense, as ncclGather / torch.distributed.gather deliver them) -> d_full[H][W][3], on `stream`.
extern "C" int rt_shard_place_rows_device(const rt_scene *sc, const rt_opts *o, int n_ranks, int pad_rows,
                                          const void *d_gathered, void *d_full, void *stream) {
    if (!sc || !d_gathered || !d_full || n_ranks < 1) {
        set_error("rt_shard_place_rows_device: null or out-of-range argument");
        return RT_ERR_ARG;
    }
    const int tile_rows = (o && o->tile_rows > 0) ? o->tile_rows : 8;
    const int rotate = (o && n_ranks > 1) ? o->tile_rotate : 0;  // how the ranks' shards were cut (rt_opts.tile_rotate)
    if (rotate < 0 || rotate > 2) {
        set_error("rt_shard_place_rows_device: tile_rotate %d", rotate);
        return RT_ERR_ARG;
    }
    const int W = sc->s.width, H = sc->s.height;
    const int tiles = (H + tile_rows - 1) / tile_rows;
    const int need = ((tiles + n_ranks - 1) / n_ranks) * tile_rows;  // rows of the largest shard, rounded up to whole tiles
    if (pad_rows < std::min(need, H)) {
        // the exact requirement is max over ranks of rt_shard_rows(); `need` over-estimates it by < tile_rows
        rt_opts probe;
        rt_opts_default(&probe);
        probe.tile_rows = tile_rows, probe.tile_stride = n_ranks, probe.tile_rotate = rotate;
        int worst = 0;
        for (int r = 0; r < n_ranks; ++r) {
            probe.tile_first = r;
            worst = std::max(worst, rt_shard_rows(sc, &probe));
        }
        if (pad_rows < worst) {
            set_error("rt_shard_place_rows_device: pad_rows %d is smaller than the largest shard (%d rows)", pad_rows, worst);
            return RT_ERR_ARG;
        }
    }
    const size_t total = (size_t)H * W * 3;
    if (!total) return RT_OK;
    hipLaunchKernelGGL(place_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float *)d_gathered, (float *)d_full, H, W * 3, tile_rows, n_ranks, pad_rows, rotate);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

// release the streams, communicators and buffers rt_render_hip_tiles keeps between calls (a group with a frame in flight
// on another thread is released when that frame ends: the frame holds a reference)
extern "C" void rt_tiles_shutdown(void) {
    std::vector<std::shared_ptr<TileGroup>> gone;
    {
        std::lock_guard<std::mutex> lock(g_groups_mu);
        gone.swap(g_groups);
    }
    for (auto &g : gone) {
        std::lock_guard<std::mutex> frame(g->mu);  // (waits for a frame in flight)
    }
    gone.clear();
}
