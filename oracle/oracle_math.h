// TEST INFRASTRUCTURE ONLY. CPU restatement (oracle) of the DHRRAM/RayTracer hot path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the product
// (raytracer_amd/, include/) never links, imports or calls it.
//
// Small fp32 vector/colour types with the operator set the reference's framework types are used with
// (cyVector/cyColor are absent from the reference repository). Every operation is a plain
// fp32 expression evaluated left to right; the file is compiled with -ffp-contract=off so no FMA is formed.
#pragma once
// powf / expf / sinf / cosf / tanf / asinf / acosf / atan2f: the routines the kernels evaluate too (stand-in for the
// platform libm the reference calls; see the header). Shared on purpose: identical source, identical results.
//
// ORC_PLATFORM_LIBM (make LIBM=platform -> liboracle_libm.so): the second leg. The same restatement, but the eight functions
// are the platform's libm, called the way the reference calls them (std::pow of the Blinn lobe, Shading.cpp:334,340;
// std::exp in ApplyAbsorption, RayCaster.cpp:281-301; std::sin / std::cos of the lens and light sampling ...), and shares no
// arithmetic source with the product. Its samples are not the kernels' bit for bit - a last bit in a powf is another path a
// few bounces on - so it is held to the product at image level: tests/test_libm_leg_*.py.
#include <cmath>
#ifdef ORC_PLATFORM_LIBM
inline float rtx_powf(float x, float y) { return std::pow(x, y); }
inline float rtx_expf(float x) { return std::exp(x); }
inline float rtx_sinf(float x) { return std::sin(x); }
inline float rtx_cosf(float x) { return std::cos(x); }
inline float rtx_tanf(float x) { return std::tan(x); }
inline float rtx_asinf(float x) { return std::asin(x); }
inline float rtx_acosf(float x) { return std::acos(x); }
inline float rtx_atan2f(float y, float x) { return std::atan2(y, x); }
#else
#include "../raytracer_amd/csrc/rtx_math.h"
#endif
#include <cstdint>
#include <algorithm>

namespace orc {

constexpr float BIGFLOAT = 1.0e30f;  // framework constant (stand-in choice, DESIGN.md "unpinned")
constexpr float PI_F = 3.14159265358979323846f;  // float(M_PI)

struct V3 {
    float x, y, z;
    V3() : x(0), y(0), z(0) {}
    V3(float a, float b, float c) : x(a), y(b), z(c) {}
    explicit V3(float a) : x(a), y(a), z(a) {}
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float& at(int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 operator+(V3 a, V3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator-(V3 a) { return V3(-a.x, -a.y, -a.z); }
inline V3 operator*(V3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(float s, V3 a) { return V3(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(V3 a, V3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V3 operator/(V3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline float len2(V3 a) { return dot(a, a); }
inline float len(V3 a) { return std::sqrt(len2(a)); }
// cy::Vec3::GetNormalized: v / Length()
inline V3 normalized(V3 a) { return a / len(a); }
inline float vmax(V3 a) { return std::max(a.x, std::max(a.y, a.z)); }
inline float vsum(V3 a) { return a.x + a.y + a.z; }
// cy::Color::Luma1 (Rec.601 weights; stand-in choice)
inline float luma1(V3 c) { return 0.299f * c.x + 0.587f * c.y + 0.114f * c.z; }
inline V3 clamp_max(V3 c, float m) { return V3(std::min(c.x, m), std::min(c.y, m), std::min(c.z, m)); }
inline V3 clamp_min(V3 c, float m) { return V3(std::max(c.x, m), std::max(c.y, m), std::max(c.z, m)); }

// RayCaster.cpp:96-100 / Shading.cpp:25-29
inline V3 safe_normalize(V3 v) {
    float l2 = len2(v);
    if (l2 <= 0.0f) return V3(0, 0, 0);
    return v / std::sqrt(l2);
}
inline float clamp01(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }

// 3x4 row-major affine transform helpers
struct M34 {
    float m[12];
};
inline V3 xf_point(const M34& M, V3 p) {
    return V3(M.m[0] * p.x + M.m[1] * p.y + M.m[2] * p.z + M.m[3], M.m[4] * p.x + M.m[5] * p.y + M.m[6] * p.z + M.m[7],
              M.m[8] * p.x + M.m[9] * p.y + M.m[10] * p.z + M.m[11]);
}
inline V3 xf_vec(const M34& M, V3 p) {
    return V3(M.m[0] * p.x + M.m[1] * p.y + M.m[2] * p.z, M.m[4] * p.x + M.m[5] * p.y + M.m[6] * p.z,
              M.m[8] * p.x + M.m[9] * p.y + M.m[10] * p.z);
}
// transpose of linear part times v (normals: N' = itm^T N)
inline V3 xf_vec_t(const M34& M, V3 p) {
    return V3(M.m[0] * p.x + M.m[4] * p.y + M.m[8] * p.z, M.m[1] * p.x + M.m[5] * p.y + M.m[9] * p.z,
              M.m[2] * p.x + M.m[6] * p.y + M.m[10] * p.z);
}
inline bool is_identity(const M34& M) {
    static const float I[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    for (int i = 0; i < 12; ++i)
        if (M.m[i] != I[i]) return false;
    return true;
}
inline M34 invert(const M34& M) {
    // double precision inverse of the linear part, then fp32
    double a = M.m[0], b = M.m[1], c = M.m[2], d = M.m[4], e = M.m[5], f = M.m[6], g = M.m[8], h = M.m[9], i = M.m[10];
    double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    double id = 1.0 / det;
    double r[9] = {(e * i - f * h) * id, (c * h - b * i) * id, (b * f - c * e) * id,
                   (f * g - d * i) * id, (a * i - c * g) * id, (c * d - a * f) * id,
                   (d * h - e * g) * id, (b * g - a * h) * id, (a * e - b * d) * id};
    M34 R;
    for (int k = 0; k < 3; ++k) {
        R.m[k * 4 + 0] = float(r[k * 3 + 0]);
        R.m[k * 4 + 1] = float(r[k * 3 + 1]);
        R.m[k * 4 + 2] = float(r[k * 3 + 2]);
        R.m[k * 4 + 3] = float(-(r[k * 3 + 0] * M.m[3] + r[k * 3 + 1] * M.m[7] + r[k * 3 + 2] * M.m[11]));
    }
    return R;
}

// ---- counter-based RNG (documented deviation: SURVEY.md §8a row R; the reference's per-thread RNG stream
// makes its output run-to-run nondeterministic, RayCaster.cpp:1178-1181,1288). ----
inline uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
// key: (a,b) = (pixel index, sample index) or (photon id, 0xFFFFFFFF); stream = bounce; dim = draw slot.
inline float rnd01(uint64_t seed, uint32_t a, uint32_t b, uint32_t stream, uint32_t dim) {
    uint64_t k1 = mix64(seed ^ ((uint64_t(a) << 32) | uint64_t(b)));
    uint64_t k2 = mix64(k1 + ((uint64_t(stream) << 32) | uint64_t(dim)) * 0x9e3779b97f4a7c15ULL + 0x9e3779b97f4a7c15ULL);
    return float(k2 >> 40) * (1.0f / 16777216.0f);
}
// draw slots
enum { DIM_LOBE = 0, DIM_U1 = 1, DIM_U2 = 2, DIM_RR = 3, DIM_LIGHT0 = 4 };

}  // namespace orc
