// The confidence-interval test of the adaptive sampler (reference RayTracer/RayCaster.cpp:848-868 / :957-975) on one
// pixel's running statistics: sum = {sum r, g, b, bits(n)}, sum2 = {sum r^2, g^2, b^2, running z}.
// Translation units that include this are compiled with IEEE division and square root (see kernels_post.hip).
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ bool ci_needs_refinement(const float4 s1, const float4 s2, float maxHalf, int sppMax) {
    const int n = __float_as_int(s1.w);
    if (n <= 1 || n >= sppMax) return false;
    const float fn = (float)n;
    const float meanR = s1.x / fn, meanG = s1.y / fn, meanB = s1.z / fn;
    const float varR = s2.x / fn - meanR * meanR;
    const float varG = s2.y / fn - meanG * meanG;
    const float varB = s2.z / fn - meanB * meanB;
    float vm = fmaxf(varR, fmaxf(varG, varB));
    if (vm < 0.0f) vm = 0.0f;
    const float stddev = sqrtf(vm);
    const int df = n - 1;
    float t;
    if (df <= 3) t = 3.182f;
    else if (df <= 7) t = 2.365f;
    else if (df <= 15) t = 2.131f;
    else if (df <= 30) t = 2.042f;
    else t = 1.96f;
    const float half_width = t * stddev / sqrtf(fn);
    return half_width > maxHalf;
}
