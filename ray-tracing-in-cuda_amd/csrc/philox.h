// Philox4x32-10 counter-based RNG (Salmon et al., SC'11), host + device.
//
// Replaces the reference's per-pixel curand XORWOW state (48 B/pixel in global
// memory, gpu-version/main.cu:120-125, 487-496) with a stream that needs no memory:
//   * every (pixel, sample) gets its own generator, seeded by ONE counter-based block
//         Philox4x32-10(counter = (pixel_id, sample_index, 0, 0), key = (seed_lo, seed_hi))
//     whose four words are the state (x, y, z, w) of
//   * Marsaglia's xorshift128 ("xor128", 7 integer ops per draw), which yields the
//     sample's uniforms in sequence.
// Samples are independent of each other and of the schedule (counter-based seeding);
// inside a sample the cheap generator avoids one Philox block per four draws, which
// every inlined call site paid on every iteration because some lane of the wave is
// always at a block boundary.  A uniform is the top 24 bits, xi = (word >> 8) * 2^-24
// in [0,1), exactly representable in fp32 and fp64, which is what lets the compiled
// fp64 reference (oracle/_ref) consume the identical stream through its rand() hook.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RTMI_HD __host__ __device__ __forceinline__
#else
#define RTMI_HD inline
#endif

namespace rtmi {

struct Philox4 {
    uint32_t v[4];
};

// 32 x 32 -> 64-bit product as ONE wide multiply (device: v_mad_u64_u32, one quarter-rate instruction
// instead of v_mul_hi_u32 + v_mul_lo_u32)
RTMI_HD void philox_mul(uint32_t a, uint32_t b, uint32_t &hi, uint32_t &lo) {
    const uint64_t p = (uint64_t)a * (uint64_t)b;
    hi = (uint32_t)(p >> 32);
    lo = (uint32_t)p;
}

RTMI_HD Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                              uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    const uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0, lo0, hi1, lo1;
        philox_mul(M0, c0, hi0, lo0);
        philox_mul(M1, c2, hi1, lo1);
        uint32_t n0 = hi1 ^ c1 ^ k0;
        uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0;
        c1 = lo1;
        c2 = n2;
        c3 = lo0;
        k0 += W0;
        k1 += W1;
    }
    Philox4 o;
    o.v[0] = c0;
    o.v[1] = c1;
    o.v[2] = c2;
    o.v[3] = c3;
    return o;
}

struct Xor128 {
    uint32_t x, y, z, w;
};

RTMI_HD Xor128 xor128_seed(uint32_t pixel, uint32_t sample, uint32_t k0, uint32_t k1) {
    Philox4 p = philox4x32_10(pixel, sample, 0u, 0u, k0, k1);
    Xor128 g;
    g.x = p.v[0], g.y = p.v[1], g.z = p.v[2], g.w = p.v[3];
    if ((g.x | g.y | g.z | g.w) == 0u) g.w = 0x9E3779B9u;  // the all-zero state is a fixed point
    return g;
}

RTMI_HD uint32_t xor128_next(Xor128 &g) {  // Marsaglia, "Xorshift RNGs" (2003), xor128
    const uint32_t t = g.x ^ (g.x << 11);
    g.x = g.y, g.y = g.z, g.z = g.w;
    g.w = g.w ^ (g.w >> 19) ^ (t ^ (t >> 8));
    return g.w;
}

}  // namespace rtmi
