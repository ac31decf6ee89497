// Post-processing on the finished frame: the 3x3 median filter of the reference's built-in denoiser
// (RayTracer/Denoiser.cpp:19-81 MedianAt / MedianFilterToLinear, :93-134 ApplyMedian's 8-bit conversion).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"

// median of nine by a partial selection network on registers (the reference's nth_element picks the same value)
__device__ __forceinline__ void mnmx(float& a, float& b) {
    const float lo = fminf(a, b), hi = fmaxf(a, b);
    a = lo;
    b = hi;
}
__device__ __forceinline__ float median9(float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7, float v8) {
    mnmx(v1, v2); mnmx(v4, v5); mnmx(v7, v8); mnmx(v0, v1); mnmx(v3, v4); mnmx(v6, v7);
    mnmx(v1, v2); mnmx(v4, v5); mnmx(v7, v8); mnmx(v0, v3); mnmx(v5, v8); mnmx(v4, v7);
    mnmx(v3, v6); mnmx(v1, v4); mnmx(v2, v5); mnmx(v4, v7); mnmx(v4, v2); mnmx(v6, v4);
    mnmx(v4, v2);
    return v4;
}

// One lane per pixel and all three channels; the 3x3 window is clamped to the image (Denoiser.cpp:25-27).
__global__ void median3_kernel(const float* __restrict__ src, int W, int H, int srgb, float* __restrict__ linear,
                               unsigned char* __restrict__ rgb8) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= W || y >= H) return;
    float v[3][9];
    int k = 0;
    for (int dy = -1; dy <= 1; ++dy) {
        const int yy = min(max(y + dy, 0), H - 1);
        for (int dx = -1; dx <= 1; ++dx) {
            const int xx = min(max(x + dx, 0), W - 1);
            const float* p = src + 3 * ((size_t)yy * W + xx);
            v[0][k] = p[0];
            v[1][k] = p[1];
            v[2][k] = p[2];
            ++k;
        }
    }
    const size_t i = (size_t)y * W + x;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float m = median9(v[c][0], v[c][1], v[c][2], v[c][3], v[c][4], v[c][5], v[c][6], v[c][7], v[c][8]);
        if (linear) linear[3 * i + c] = m;
        if (rgb8) {   // Denoiser.cpp:108-118: clamp, optional sRGB, truncate
            m = fminf(fmaxf(m, 0.0f), 1.0f);
            if (srgb) m = m < 0.0031308f ? m * 12.92f : powf(m, 0.41666f) * 1.055f - 0.055f;
            rgb8[3 * i + c] = (unsigned char)fminf(fmaxf(m * 255.0f, 0.0f), 255.0f);
        }
    }
}

void launch_median3(const float* src, int W, int H, int srgb, float* linear, unsigned char* rgb8, hipStream_t stream) {
    dim3 block(64, 4), grid((W + 63) / 64, (H + 3) / 4);
    hipLaunchKernelGGL(median3_kernel, grid, block, 0, stream, src, W, H, srgb, linear, rgb8);
}
