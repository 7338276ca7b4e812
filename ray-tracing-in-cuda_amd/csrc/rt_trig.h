// fp32 atan2 / acos with a FIXED operation sequence (every fused multiply-add explicit), for the hit record's
// texture coordinates (sphere: gpu-version/object.cuh:87-93, cylinder: :283-288).  libm and the device math
// library round differently, and the framebuffer of the HIP path is held to bit-equality with the CPU checker,
// which restates this sequence operation for operation.  Accuracy: the
// argument reduction and polynomial of Cephes' atanf (~2 ulp), far below a texel of any texture.
#pragma once
#include <math.h>

#ifndef RTMI_HD
#ifdef __HIPCC__
#define RTMI_HD __host__ __device__ inline
#else
#define RTMI_HD inline
#endif
#endif

namespace rtmi {

RTMI_HD float rt_atan_unit(float a) {  // atan(a), 0 <= a <= 1
    float y0 = 0.0f, t = a;
    if (a > 0.4142135679721832275390625f) {  // tan(pi / 8): atan(a) = pi/4 + atan((a - 1) / (a + 1))
        y0 = 0.785398185253143310546875f;
        t = (a - 1.0f) / (a + 1.0f);
    }
    const float z = t * t;
    float p = fmaf(8.05374449538e-2f, z, -1.38776856032e-1f);
    p = fmaf(p, z, 1.99777106478e-1f);
    p = fmaf(p, z, -3.33329491539e-1f);
    return y0 + fmaf(p * z, t, t);
}

RTMI_HD float rt_atan2f(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    float r = 0.0f;
    if (mx > 0.0f) r = rt_atan_unit(mn / mx);
    if (ay > ax) r = 1.57079637050628662109375f - r;
    if (x < 0.0f) r = 3.1415927410125732421875f - r;
    return y < 0.0f ? -r : r;
}

RTMI_HD float rt_acosf(float c) {  // -1 <= c <= 1
    const float s = sqrtf(fmaxf((1.0f - c) * (1.0f + c), 0.0f));
    return rt_atan2f(s, c);
}

}  // namespace rtmi
