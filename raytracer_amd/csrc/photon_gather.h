// Device-side photon gather: EstimateCaustics -> PhotonMap::EstimateIrradiance<N, PHOTONMAP_FILTER_LINEAR>
// (reference RayTracer/RayCaster.cpp:143-153; cyPhotonMap itself is absent from the reference).
#pragma once
#include "pathstate.h"
#include "vec.h"
#ifndef PI_F
#define PI_F 3.14159265358979323846f
#endif

// ---- photon estimate ------------------------------------------------------------------------------------------
// k nearest photons within `radius` via a uniform grid; see DESIGN.md for the definition shared with the oracle.
// Two passes: (1) the k-th smallest (squared distance, photon key) by bisection on the 64-bit key, (2) the cone-filtered
// sum over the photons up to it, in storage order = (cell, photon key) order. The oracle sums in the same order over the
// same lattice (anchored at the origin, edge chosen from the photons' own bounding box), so estimates agree bit for bit.
__device__ inline f3 estimate_irradiance(const DevPhotonMap& pm, f3 pos, f3 normal) {
    if (pm.n <= 0) return mk3(0.0f);
    const float radius = pm.radius;
    const float r2 = radius * radius;
    const int K = pm.maxPhotons;
    // cells of the lattice floor(p / cell) that the search sphere's box touches, clamped to the grid window (in float, so
    // that a far-away query cannot overflow the conversion); the distance test decides
    const auto lc = [&](float x, int a, int g) { return (int)fminf(fmaxf(floorf(x / pm.cell) - (float)pm.imin[a], 0.0f), (float)(g - 1)); };
    const int cx0 = lc(pos.x - radius, 0, pm.gx), cx1 = lc(pos.x + radius, 0, pm.gx);
    const int cy0 = lc(pos.y - radius, 1, pm.gy), cy1 = lc(pos.y + radius, 1, pm.gy);
    const int cz0 = lc(pos.z - radius, 2, pm.gz), cz1 = lc(pos.z + radius, 2, pm.gz);
    // pass 1: count candidates and find the K-th smallest (d2, index) by repeated threshold refinement:
    // a max-heap of K entries would need K registers; instead bisect on the key, which is exact because keys
    // are (d2 bits, index) pairs compared as 64-bit integers.
    int total = 0;
    for (int cz = cz0; cz <= cz1; ++cz)
        for (int cy = cy0; cy <= cy1; ++cy)
            for (int cx = cx0; cx <= cx1; ++cx) {
                const size_t c = ((size_t)cz * pm.gy + cy) * pm.gx + cx;
                for (uint32_t k = pm.cellStart[c]; k < pm.cellStart[c + 1]; ++k) {
                    const float4 p = pm.pos[k];
                    const f3 d = mk3(p.x, p.y, p.z) - pos;
                    if (len2_3(d) <= r2) ++total;
                }
            }
    if (total == 0) return mk3(0.0f);
    unsigned long long thr = ~0ull;   // keys <= thr are the selected photons
    float rr2 = r2;
    if (total >= K) {
        // smallest key such that count(keys <= key) >= K, by bisection over the 64-bit key space
        unsigned long long lo = 0, hi = (((unsigned long long)__float_as_uint(r2)) << 32) | 0xFFFFFFFFull;
        while (lo < hi) {
            const unsigned long long mid = lo + ((hi - lo) >> 1);
            int cnt = 0;
            for (int cz = cz0; cz <= cz1; ++cz)
                for (int cy = cy0; cy <= cy1; ++cy)
                    for (int cx = cx0; cx <= cx1; ++cx) {
                        const size_t c = ((size_t)cz * pm.gy + cy) * pm.gx + cx;
                        for (uint32_t k = pm.cellStart[c]; k < pm.cellStart[c + 1]; ++k) {
                            const float4 p = pm.pos[k];
                            const f3 d = mk3(p.x, p.y, p.z) - pos;
                            const float d2 = len2_3(d);
                            if (d2 <= r2) {
                                const unsigned long long key = (((unsigned long long)__float_as_uint(d2)) << 32) | (unsigned long long)__float_as_uint(p.w);
                                if (key <= mid) ++cnt;
                            }
                        }
                    }
            if (cnt >= K) hi = mid;
            else lo = mid + 1;
        }
        thr = lo;
        rr2 = __uint_as_float((uint32_t)(thr >> 32));
    }
    if (rr2 <= 0.0f) return mk3(0.0f);
    const float r = sqrtf(rr2);
    f3 irrad = mk3(0.0f);
    // pass 2: the sum, in grid order
    for (int cz = cz0; cz <= cz1; ++cz)
        for (int cy = cy0; cy <= cy1; ++cy)
            for (int cx = cx0; cx <= cx1; ++cx) {
                const size_t c = ((size_t)cz * pm.gy + cy) * pm.gx + cx;
                for (uint32_t k = pm.cellStart[c]; k < pm.cellStart[c + 1]; ++k) {
                    const float4 p = pm.pos[k];
                    const f3 d = mk3(p.x, p.y, p.z) - pos;
                    const float d2 = len2_3(d);
                    if (d2 > r2) continue;
                    const unsigned long long key = (((unsigned long long)__float_as_uint(d2)) << 32) | (unsigned long long)__float_as_uint(p.w);
                    if (key > thr) continue;
                    const float4 dd = pm.dir[k];
                    if (dot3(mk3(dd.x, dd.y, dd.z), normal) >= 0.0f) continue;
                    const float w = 1.0f - sqrtf(d2) / r;
                    const float4 pw = pm.power[k];
                    irrad = irrad + mk3(pw.x, pw.y, pw.z) * w;
                }
            }
    return irrad * (3.0f / (PI_F * rr2));
}

