// Minimal fp32 3-vector for device code. Every operation is a plain component-wise fp32 expression; functions
// that must reproduce an fp32 CPU evaluation bit for bit switch contraction off locally.
#pragma once
#include <hip/hip_runtime.h>
#include "rtx_math.h"   // powf, sinf ... of the path: the same routines the CPU oracle evaluates

struct f3 {
    float x, y, z;
};
__device__ __forceinline__ f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
__device__ __forceinline__ f3 mk3(float a) { return f3{a, a, a}; }
__device__ __forceinline__ f3 ld3(const float* p) { return f3{p[0], p[1], p[2]}; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f3 operator-(f3 a) { return f3{-a.x, -a.y, -a.z}; }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ f3 operator*(float s, f3 a) { return f3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return f3{a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ f3 operator/(f3 a, float s) { return f3{a.x / s, a.y / s, a.z / s}; }
__device__ __forceinline__ float dot3(f3 a, f3 b) {
#pragma clang fp contract(off)
    return a.x * b.x + a.y * b.y + a.z * b.z;
}
__device__ __forceinline__ f3 cross3(f3 a, f3 b) {
#pragma clang fp contract(off)
    return f3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ float len2_3(f3 a) { return dot3(a, a); }
__device__ __forceinline__ float len3(f3 a) { return sqrtf(len2_3(a)); }
__device__ __forceinline__ f3 normalized3(f3 a) { return a / len3(a); }
__device__ __forceinline__ f3 safe_normalize3(f3 v) {
    const float l2 = len2_3(v);
    if (l2 <= 0.0f) return mk3(0.0f);
    return v / sqrtf(l2);
}
__device__ __forceinline__ float max3c(f3 a) { return fmaxf(a.x, fmaxf(a.y, a.z)); }
__device__ __forceinline__ float luma1(f3 c) {
#pragma clang fp contract(off)
    return 0.299f * c.x + 0.587f * c.y + 0.114f * c.z;
}
__device__ __forceinline__ f3 clamp_max3(f3 c, float m) { return f3{fminf(c.x, m), fminf(c.y, m), fminf(c.z, m)}; }
__device__ __forceinline__ float clamp01f(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }

// counter-based RNG shared by every sampling decision: 24-bit uniform in [0,1).
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
__device__ __forceinline__ unsigned long long rng_key(unsigned long long seed, uint32_t a, uint32_t b) {
    return mix64(seed ^ (((unsigned long long)a << 32) | (unsigned long long)b));
}
__device__ __forceinline__ float rng_draw(unsigned long long key, uint32_t stream, uint32_t dim) {
#ifdef RTX_ABLATE_RNG   // (timing-only variant: a 32-bit hash with two multiplies; wrong images)
    uint32_t h = (uint32_t)key ^ (uint32_t)(key >> 32) ^ (stream * 0x9e3779b9u + dim * 0x85ebca6bu);
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    return (float)(h >> 8) * (1.0f / 16777216.0f);
#endif
    const unsigned long long k2 =
        mix64(key + (((unsigned long long)stream << 32) | (unsigned long long)dim) * 0x9e3779b97f4a7c15ULL + 0x9e3779b97f4a7c15ULL);
    return (float)(k2 >> 40) * (1.0f / 16777216.0f);
}
#define DIM_LOBE 0
#define DIM_U1 1
#define DIM_U2 2
#define DIM_RR 3
#define DIM_LIGHT0 4
