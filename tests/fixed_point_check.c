/* The kernel's fp64-free sample -> 64-bit fixed point conversion (render_kernel.hip, radiance_to_fixed)
 * against the checker's definition llrint((double)v * 2^24) (oracle/rt_oracle.c, radiance_to_fixed).
 * usage: fixed_point_check <stride>   (stride 1 = every fp32 bit pattern, ~20 s) */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static uint64_t by_definition(float v) {
    if (!(fabsf(v) <= 65536.0f)) v = (v != v) ? 0.0f : copysignf(65536.0f, v);
    return (uint64_t)llrint((double)v * 16777216.0);
}

/* the device expression, operation for operation (v_cvt_u32_f32 truncates, v_rndne_f32 = rintf) */
static uint64_t as_in_the_kernel(float v) {
    if (fabsf(v) < 128.0f) return (uint64_t)(int64_t)(int32_t)rintf(v * 16777216.0f); /* the fast path */
    if (!(fabsf(v) <= 65536.0f)) v = (v != v) ? 0.0f : copysignf(65536.0f, v);
    const float a = fabsf(v);
    const uint32_t hi = (uint32_t)a;
    const float frac = a - (float)hi;
    const uint32_t lo = (uint32_t)rintf(frac * 16777216.0f);
    const uint64_t m = ((uint64_t)hi << 24) + lo;
    return v < 0.0f ? 0ull - m : m;
}

int main(int argc, char **argv) {
    const uint64_t stride = argc > 1 ? strtoull(argv[1], 0, 0) : 1;
    uint64_t bad = 0, n = 0;
    for (uint64_t b = 0; b <= 0xffffffffull; b += stride) {
        uint32_t u = (uint32_t)b;
        float v;
        memcpy(&v, &u, 4);
        if (by_definition(v) != as_in_the_kernel(v)) {
            if (bad < 5) printf("mismatch %08x %g\n", u, v);
            ++bad;
        }
        ++n;
    }
    /* neighbourhoods of every power of two, of the clamp (2^16) and of ties at the 2^-25 boundary */
    for (int e = -40; e <= 31; ++e)
        for (int k = -4; k <= 4; ++k) {
            float v = ldexpf(1.0f, e);
            for (int s = 0; s < (k < 0 ? -k : k); ++s) v = nextafterf(v, k < 0 ? 0.0f : INFINITY);
            const float c[6] = {v, -v, v * 1.5f, -v * 1.5f, v * 1.25f, v * 0.75f};
            for (int i = 0; i < 6; ++i, ++n)
                if (by_definition(c[i]) != as_in_the_kernel(c[i])) ++bad;
        }
    printf("checked %llu values, %llu mismatches\n", (unsigned long long)n, (unsigned long long)bad);
    return bad != 0;
}
