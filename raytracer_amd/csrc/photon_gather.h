// Device-side photon gather: EstimateCaustics -> PhotonMap::EstimateIrradiance<N, PHOTONMAP_FILTER_LINEAR>
// (reference RayTracer/RayCaster.cpp:143-153; cyPhotonMap itself is absent from the reference).
#pragma once
#include "pathstate.h"
#include "vec.h"
#ifndef PI_F
#define PI_F 3.14159265358979323846f
#endif

// ---- photon estimate ------------------------------------------------------------------------------------------
// k nearest photons within `radius` via the uniform hash grid (cells of edge `radius`); see DESIGN.md for the
// definition shared with the oracle. Two passes: (1) the k-th smallest squared distance by a bounded max-heap
// kept in registers/scratch, (2) the cone-filtered sum over photons within it.
__device__ inline f3 estimate_irradiance(const DevPhotonMap& pm, f3 pos, f3 normal) {
    if (pm.n <= 0) return mk3(0.0f);
    const float radius = pm.radius;
    const float r2 = radius * radius;
    const int K = pm.maxPhotons;
    int cx0 = (int)floorf((pos.x - radius - pm.gmin[0]) / pm.cell), cx1 = (int)floorf((pos.x + radius - pm.gmin[0]) / pm.cell);
    int cy0 = (int)floorf((pos.y - radius - pm.gmin[1]) / pm.cell), cy1 = (int)floorf((pos.y + radius - pm.gmin[1]) / pm.cell);
    int cz0 = (int)floorf((pos.z - radius - pm.gmin[2]) / pm.cell), cz1 = (int)floorf((pos.z + radius - pm.gmin[2]) / pm.cell);
    // cell_of clamps photons into the border cells, so a query beyond the grid (photons handed in through
    // rtx_photon_set need not lie inside the scene box) searches the border cells too; the distance test decides
    cx0 = min(max(cx0, 0), pm.gx - 1); cy0 = min(max(cy0, 0), pm.gy - 1); cz0 = min(max(cz0, 0), pm.gz - 1);
    cx1 = max(min(cx1, pm.gx - 1), 0); cy1 = max(min(cy1, pm.gy - 1), 0); cz1 = max(min(cz1, pm.gz - 1), 0);
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
    // pass 2: accumulate in (d2, index) order is not reproducible on a grid walk; sum in grid order instead
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

