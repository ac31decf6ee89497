// What is computed from the accumulated film and pixel statistics: the confidence-interval test of the adaptive
// sampler, the film resolve, and the 3x3 median filter of the reference's built-in denoiser (RayTracer/Denoiser.cpp:19-81
// MedianAt / MedianFilterToLinear, :93-134 ApplyMedian's 8-bit conversion).
// This translation unit is compiled with IEEE division and square root (kernels_path.hip is not): the variance
// s2/n - (s1/n)^2 of a pixel whose samples are all equal must come out as exactly zero, as it does on the CPU - with a
// 2.5-ulp quotient a constant bright pixel (a lamp clamped to 10) shows a spurious deviation above the 0.005 threshold
// and is refined for nothing.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"
#include "rtx_math.h"
#include "pathstate.h"
#include "adaptive.h"

__device__ __forceinline__ uint32_t wave_append(bool pred, uint32_t* counter) {
    const unsigned long long mask = __ballot(pred);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t total = (uint32_t)__popcll(mask);
    uint32_t base = 0;
    if (total) {
        const int leader = __ffsll((long long)mask) - 1;
        if ((int)lane == leader) base = atomicAdd(counter, total);
        base = __shfl(base, leader);
    }
    const uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    return base + prefix;
}


// Confidence-interval test (RayCaster.cpp:848-868 / :957-975): pixels that still need samples go to the next list.
__global__ void adaptive_kernel(PathPool pool, const int* __restrict__ pixels, uint32_t nPix, float maxHalf, int sppMax,
                                int* __restrict__ pixelsNext, uint32_t* counter) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool refine = false;
    int pix = -1;
    if (i < nPix) {
        pix = pixels ? pixels[i] : (int)i;   // no list: the whole frame
        if (pix >= 0) {
            refine = ci_needs_refinement(pool.statSum[pix], pool.statSum2[pix], maxHalf, sppMax);
        }
    }
    const uint32_t q = wave_append(refine, counter);
    if (refine) pixelsNext[q] = pix;
}


// mergeToImage + ToColor24 (RayCaster.cpp:1188-1211, 33-48); Linear2sRGB of the absent cyColor: stand-in
// c < 0.0031308 ? 12.92 c : 1.055 c^0.41666 - 0.055
__global__ void film_resolve_kernel(rtx_film film, uint32_t n, int srgb, float* __restrict__ linear, unsigned char* __restrict__ rgb8) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 a = reinterpret_cast<const float4*>(film.accum)[i];
    float c[3] = {0.0f, 0.0f, 0.0f};
    if (a.w > 0.0f) {
        c[0] = a.x / a.w;
        c[1] = a.y / a.w;
        c[2] = a.z / a.w;
    }
    for (int k = 0; k < 3; ++k) {
        if (linear) linear[3 * (size_t)i + k] = c[k];
        float x = c[k];
        if (srgb) x = x < 0.0031308f ? x * 12.92f : rtx_powf(x, 0.41666f) * 1.055f - 0.055f;
        x = fmaxf(x, 0.0f);
        x = fminf(x, 1.0f);
        if (rgb8) rgb8[3 * (size_t)i + k] = (unsigned char)(x * 255.0f);
    }
}


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
            if (srgb) m = m < 0.0031308f ? m * 12.92f : rtx_powf(m, 0.41666f) * 1.055f - 0.055f;
            rgb8[3 * i + c] = (unsigned char)fminf(fmaxf(m * 255.0f, 0.0f), 255.0f);
        }
    }
}

// mergeToImage across GPUs (RayCaster.cpp:1182-1212): dst += the films of the other group members, in member order
// (a fixed order of float additions: the merged frame does not depend on which member finished first), z by min;
// the first-hit guides of a pixel are written by one member only, so they add up as well.
struct FilmMergeSrc {
    rtx_film f[RTX_GROUP_MAX - 1];
    int n;
};
__global__ void film_merge_kernel(rtx_film dst, FilmMergeSrc src, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 a = reinterpret_cast<float4*>(dst.accum)[i];
    int c = dst.count[i];
    float z = dst.z[i];
    float4 al = dst.albedo ? reinterpret_cast<float4*>(dst.albedo)[i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float4 nn = dst.normal ? reinterpret_cast<float4*>(dst.normal)[i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (int k = 0; k < src.n; ++k) {
        const rtx_film& s = src.f[k];
        const float4 b = reinterpret_cast<const float4*>(s.accum)[i];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        c += s.count[i];
        z = fminf(z, s.z[i]);
        if (dst.albedo && s.albedo) {
            const float4 t = reinterpret_cast<const float4*>(s.albedo)[i];
            al.x += t.x; al.y += t.y; al.z += t.z; al.w += t.w;
        }
        if (dst.normal && s.normal) {
            const float4 t = reinterpret_cast<const float4*>(s.normal)[i];
            nn.x += t.x; nn.y += t.y; nn.z += t.z; nn.w += t.w;
        }
    }
    reinterpret_cast<float4*>(dst.accum)[i] = a;
    dst.count[i] = c;
    dst.z[i] = z;
    if (dst.albedo) reinterpret_cast<float4*>(dst.albedo)[i] = al;
    if (dst.normal) reinterpret_cast<float4*>(dst.normal)[i] = nn;
}
void launch_film_merge(const rtx_film& dst, const rtx_film* srcs, int nSrc, uint32_t n, hipStream_t stream) {
    if (!n || nSrc <= 0) return;
    FilmMergeSrc m;
    m.n = nSrc;
    for (int k = 0; k < nSrc; ++k) m.f[k] = srcs[k];
    hipLaunchKernelGGL(film_merge_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, dst, m, n);
}

// Adaptive rounds over several GPUs (rtx_group.cpp): the statistics of the pixels one member sampled in the round just
// finished replace the root's copies. Every listed pixel is sampled by exactly one member per round, so this is a plain
// take, not a sum.
__global__ void stats_take_kernel(float4* __restrict__ dstSum, float4* __restrict__ dstSum2, const float4* __restrict__ srcSum,
                                  const float4* __restrict__ srcSum2, const int* __restrict__ pixels, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int p = pixels[i];
    dstSum[p] = srcSum[p];
    dstSum2[p] = srcSum2[p];
}
void launch_stats_take(float* dstSum, float* dstSum2, const float* srcSum, const float* srcSum2, const int* pixels, uint32_t n,
                       hipStream_t stream) {
    if (!n) return;
    hipLaunchKernelGGL(stats_take_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, (float4*)dstSum, (float4*)dstSum2, (const float4*)srcSum,
                       (const float4*)srcSum2, pixels, n);
}

void launch_median3(const float* src, int W, int H, int srgb, float* linear, unsigned char* rgb8, hipStream_t stream) {
    dim3 block(64, 4), grid((W + 63) / 64, (H + 3) / 4);
    hipLaunchKernelGGL(median3_kernel, grid, block, 0, stream, src, W, H, srgb, linear, rgb8);
}
void launch_adaptive(const PathPool& pool, const int* pixels, uint32_t nPix, float maxHalf, int sppMax, int* pixelsNext,
                     uint32_t* counter, hipStream_t stream) {
    if (!nPix) return;
    hipLaunchKernelGGL(adaptive_kernel, dim3((nPix + 255) / 256), dim3(256), 0, stream, pool, pixels, nPix, maxHalf, sppMax,
                       pixelsNext, counter);
}
void launch_film_resolve(const rtx_film& film, uint32_t n, int srgb, float* linear, unsigned char* rgb8, hipStream_t stream) {
    hipLaunchKernelGGL(film_resolve_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, film, n, srgb, linear, rgb8);
}

// rtx_debug_math: the shared elementary functions (rtx_math.h) evaluated on the device
__global__ void math_probe_kernel(int fn, const float* __restrict__ x, const float* __restrict__ y, uint32_t n, float* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r;
    switch (fn) {
        case 0: r = rtx_powf(x[i], y[i]); break;
        case 1: r = rtx_expf(x[i]); break;
        case 2: r = rtx_sinf(x[i]); break;
        case 3: r = rtx_cosf(x[i]); break;
        case 4: r = rtx_tanf(x[i]); break;
        case 5: r = rtx_asinf(x[i]); break;
        case 6: r = rtx_acosf(x[i]); break;
        case 8: r = rtx_unorm8(x[i]); break;
        default: r = rtx_atan2f(x[i], y[i]); break;
    }
    out[i] = r;
}
void launch_math_probe(int fn, const float* x, const float* y, uint32_t n, float* out, hipStream_t stream) {
    if (!n) return;
    hipLaunchKernelGGL(math_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, fn, x, y, n, out);
}
