// LDS-broadcast vs VALU probe (measurement aid for DESIGN.md, not part of the product).
// A wave scans N spheres from LDS (one ds_read_b128 per sphere, all lanes the same address) and runs
// the 12-instruction miss path of the sphere test for RAYS rays per lane.  Reports time per
// wave-level sphere visit and per ray test at several occupancies.
//   build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o lds_probe lds_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int RAYS, int WAVES>
__global__ __launch_bounds__(256, WAVES) void scan_kernel(const float4 *spheres, int n, int reps, float *out) {
    extern __shared__ float4 lds[];
    for (int i = threadIdx.x; i <= n; i += 256) lds[i] = spheres[i < n ? i : 0];
    __syncthreads();
    float ox[RAYS], oy[RAYS], oz[RAYS], dx[RAYS], dy[RAYS], dz[RAYS], a[RAYS], best[RAYS];
    for (int r = 0; r < RAYS; ++r) {
        ox[r] = threadIdx.x * 0.01f + r, oy[r] = 100.0f + r, oz[r] = blockIdx.x * 1e-3f;
        dx[r] = 0.1f * r, dy[r] = 1.0f, dz[r] = 0.01f * threadIdx.x;
        a[r] = fmaf(dx[r], dx[r], fmaf(dy[r], dy[r], dz[r] * dz[r]));
        best[r] = 1e30f;
    }
    for (int rep = 0; rep < reps; ++rep) {
        float4 nxt = lds[0];
#pragma unroll 4
        for (int i = 0; i < n; ++i) {
            const float4 s = nxt;
            nxt = lds[i + 1];  // software prefetch (the table has one pad entry)
#pragma unroll
            for (int r = 0; r < RAYS; ++r) {
                const float cx = ox[r] - s.x, cy = oy[r] - s.y, cz = oz[r] - s.z;
                const float hb = fmaf(cx, dx[r], fmaf(cy, dy[r], cz * dz[r]));
                const float cc = fmaf(cx, cx, fmaf(cy, cy, cz * cz)) - s.w;
                const float disc = fmaf(hb, hb, -(a[r] * cc));
                if (__builtin_expect(!(disc < 0.0f) && !(hb >= 0.0f && cc >= 0.0f), 0)) {
                    const float t = (-hb - sqrtf(disc)) / a[r];
                    if (t < best[r]) best[r] = t;
                }
            }
        }
        for (int r = 0; r < RAYS; ++r) oy[r] += 1.0f;  // keep the loop body from being hoisted
    }
    float sum = 0;
    for (int r = 0; r < RAYS; ++r) sum += best[r] + oy[r];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
}

template <int RAYS, int WAVES>
static void run(const float4 *d_s, int n, int reps, float *d_out, int cus) {
    const int blocks = cus * WAVES;  // 4 waves per block -> WAVES waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    scan_kernel<RAYS, WAVES><<<blocks, 256, (n + 1) * 16>>>(d_s, n, 2, d_out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    scan_kernel<RAYS, WAVES><<<blocks, 256, (n + 1) * 16>>>(d_s, n, reps, d_out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double visits_per_simd = (double)WAVES * n * reps;  // wave-level sphere visits per SIMD
    const double ns_visit = ms * 1e6 / visits_per_simd;
    printf("rays/lane %d waves/SIMD %d: %.2f ms  %.2f ns per wave visit per SIMD (%.1f cyc @2.2GHz), "
           "%.2f ns per wave64 ray test (%.1f cyc)\n",
           RAYS, WAVES, ms, ns_visit, ns_visit * 2.2, ns_visit / RAYS, ns_visit * 2.2 / RAYS);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int n = 512;
    std::vector<float4> h(n);
    for (int i = 0; i < n; ++i) h[i] = make_float4(i * 3.0f, -50.0f - i, i * 0.5f, 0.04f);  // all misses
    float4 *d_s;
    float *d_out;
    hipMalloc(&d_s, n * 16);
    hipMalloc(&d_out, (size_t)p.multiProcessorCount * 8 * 256 * 4);
    hipMemcpy(d_s, h.data(), n * 16, hipMemcpyHostToDevice);
    const int cus = p.multiProcessorCount;
    run<1, 2>(d_s, n, 2000, d_out, cus);
    run<1, 4>(d_s, n, 1000, d_out, cus);
    run<1, 6>(d_s, n, 1000, d_out, cus);
    run<1, 8>(d_s, n, 1000, d_out, cus);
    run<2, 2>(d_s, n, 1000, d_out, cus);
    run<2, 4>(d_s, n, 1000, d_out, cus);
    run<2, 6>(d_s, n, 500, d_out, cus);
    run<4, 2>(d_s, n, 500, d_out, cus);
    run<4, 4>(d_s, n, 500, d_out, cus);
    return 0;
}
