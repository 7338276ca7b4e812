// Shader-clock and VALU-issue probe for MI355X (measurement aid for DESIGN.md, not part of the product).
//   build: hipcc --offload-arch=gfx950 -O2 -o clock_probe clock_probe.hip
// Every wave runs a chain of dependent and independent v_fma_f32 and reads s_memtime (shader clock
// counter) and s_memrealtime (constant 100 MHz) before and after.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int ILP>
__global__ __launch_bounds__(256) void fma_kernel(float *out, unsigned long long *clk, int iters) {
    float a[ILP];
    for (int i = 0; i < ILP; ++i) a[i] = threadIdx.x * 1e-3f + i;
    const float m = 1.0000001f, c = 1e-7f;
    unsigned long long t0 = __builtin_readcyclecounter();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k)
#pragma unroll
            for (int i = 0; i < ILP; ++i) a[i] = __builtin_fmaf(a[i], m, c);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < ILP; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        clk[blockIdx.x * 2 + 0] = t1 - t0;
        clk[blockIdx.x * 2 + 1] = r1 - r0;
    }
}

template <int ILP>
static void run(int blocks_per_cu, int iters) {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * blocks_per_cu;
    float *out;
    unsigned long long *clk;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipMalloc(&clk, (size_t)blocks * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    fma_kernel<ILP><<<blocks, 256>>>(out, clk, 16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    fma_kernel<ILP><<<blocks, 256>>>(out, clk, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 2);
    hipMemcpy(h.data(), clk, blocks * 16, hipMemcpyDeviceToHost);
    double ct = 0, rt = 0;
    for (int b = 0; b < blocks; ++b) ct += h[2 * b], rt += h[2 * b + 1];
    ct /= blocks, rt /= blocks;
    const double valu_per_wave = (double)iters * 16 * ILP;
    const double waves_per_simd = blocks_per_cu;  // 4 waves per block, 4 SIMDs per CU
    const double sec = rt / 100e6;
    printf("ILP %d waves/SIMD %d: %.2f ms, memtime/memrealtime = %.3f (x100 MHz = counter MHz), "
           "%.3f ns per VALU per SIMD, %.1f TFLOP/s\n",
           ILP, blocks_per_cu, ms, ct / rt, sec * 1e9 / (valu_per_wave * waves_per_simd),
           2.0 * 64 * valu_per_wave * blocks * 4 / (ms * 1e-3) / 1e12);
    hipFree(out);
    hipFree(clk);
}

int main() {
    run<1>(1, 200000);
    run<4>(1, 50000);
    run<8>(1, 25000);
    run<1>(4, 50000);
    run<4>(2, 50000);
    run<4>(6, 20000);
    run<8>(8, 10000);
    run<8>(8, 100000);
    return 0;
}
