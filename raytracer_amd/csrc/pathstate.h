// Wavefront path-tracing state in HBM (structure of arrays, one entry per path slot) and the ray queues that
// connect the kernels ray-gen -> extend -> shade -> shadow -> resolve -> film.
//
// A pass traces S samples of P pixels: slot = (chunk * S + s) * 64 + lane with pixel = pixels[chunk * 64 + lane],
// so the 64 lanes of a wave hold the same sample index of 64 neighbouring pixels (a 16x4 block of one bucket).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "layout.h"
#include "segqueue.h"

#define PF_LAST_DELTA 1u
#define PF_INSIDE_MEDIUM 2u
#define PF_HAS_PREV 4u

struct DevPhotonMap {
    const float4* pos;      // xyz, w = photon key    (sorted by cell, then key)
    const float4* dir;      // xyz
    const float4* power;    // rgb, already scaled
    const uint32_t* cellStart;  // gx*gy*gz + 1
    int imin[3];            // the grid is the window [imin, imin + g) of the lattice floor(p / cell)
    float cell;
    int gx, gy, gz;
    int n;
    float radius;
    int maxPhotons;         // 30 / 50 / 100 (RayCaster.cpp:149-151)
};

#define RTX_CULL_CAP 65536u   // entries of PathPool::cullPaths / cullShadows (a 1080p x 128 spp frame holds about 60)

struct PathPool {
    uint32_t capacity;      // slots: baseCap of the pass proper + the compaction area behind them
    // Compaction of the path state inside a pass (rtx_render.cpp "repack"): a path keeps its slot for life, so once most paths
    // of a pass have ended the survivors' 16-byte fields lie one to a 64-byte memory sector and a late bounce moves four times
    // the bytes it uses. At a few bounces of a pass shade_kernel therefore writes the state of the paths it shades to fresh,
    // consecutive slots behind the pass's own (slot >= baseCap) and the path lives on there; origin[] names the slot it was
    // generated in - the key of its pixel, its sample index and its place in the film - and unpack_kernel returns the final
    // radiance there when the pass ends.
    uint32_t baseCap;       // slots of the pass: slot = (chunk * S + s) * 64 + lane
    uint32_t* origin;       // [slot - baseCap] -> slot of the pass (0xFFFFFFFF: entry not in use, or moved on to a later area)
    // per slot
    float4* rays;           // 2 per slot: {o.xyz, tmax}, {d.xyz, 0}
    float4* hits;           // {t, u, v, bits(slot of triangle / light code)}
    float4* L;              // radiance rgb, w = primaryDist
    float4* beta;           // throughput rgb, w = lastPdf
    float4* medium;         // absorption rgb of the medium a path is inside of (PF_INSIDE_MEDIUM; the flags travel in rays[2*slot+1].w)
    float4* prevP;          // previous hit point (SamplerInfo of the previous bounce), xyz
    float4* pendBeta;       // throughput at the NEE of the current bounce
    float4* nee;            // [light][slot]: contribution rgb, w = transmittance (-1: no shadow ray)
    float4* caustic;        // kd/pi * irradiance * boost, w = valid      (only with photons)
    float4* albedo;         // first-hit albedo, w = hit flag
    float4* normal;         // first-hit geometric normal
    float2* samplePos;      // film position of the sample
    // queues
    // queues (segmented, segqueue.h): items of the two path queues, entries of the shadow queue
    uint32_t* queueA;
    uint32_t* queueB;
    float4* shadowQ;        // 2 per entry: {offset origin xyz, maxDist}, {unit dir xyz, bits(slot*8+light)}
    float4* neeQ;           // 2 per entry: {direction to the sampled point on the light, its distance}, {bits(vertex index of the bounce), bits(path slot), bits(flags | light << 8 | material << 16), alpha} (nee_ctx.h)
    SegQ qA, qB, qS, qN;    // control words of queueA / queueB / shadowQ / neeQ
    size_t neeStride;       // vertices the record array is laid out for (word k of vertex i at neeCtx[k * neeStride + i])
    float4* neeCtx;         // NEE_CTX_VEC per vertex of the current bounce, indexed by queue position (nee_ctx.h): shade_kernel -> nee_kernel
    uint32_t* totals;       // [0] entries of the queue shade appended to, [1] entries of the shadow queue,
                            // [4] entries of cullPaths, [5] entries of cullShadows
    // Box culls of the reference in scenes of world-space triangles (traverse.h ref_box_hit): the lean
    // kernels trace without them; the few hits the reference would not have reached are found afterwards (shade_kernel,
    // shadow_kernel) and handed to the general kernels. nullptr: not checked (RTX_AMD_BOX_CULLS=0, general scenes).
    uint32_t* cullPaths;    // path slots whose vertex was not shaded: their ray is traced again by the general tail kernel
    uint32_t* cullShadows;  // shadow queue entries whose chain is traced again by the general shadow kernel
    // Blocker cache of the shadow kernel (kernels_path.hip): for a cell of shadow-ray origins and a light, the opaque triangle that
    // last blocked such a ray. A hint only - whatever it holds, the kernel's answers are the same - so it is read and written
    // without synchronisation and kept across launches and frames. nullptr: switched off.
    int* blockerCache;      // blockerMask + 1 entries, -1 = empty
    uint32_t blockerMask;
    float blockerScale;     // cells per unit length
    uint32_t* counters;     // scratch counters ([8] adaptive refinement list)
    // per pixel (full frame)
    int* pixels;            // pixel list of the current round (padded to 64 with -1)
    int* pixelsNext;
    float4* statSum;        // sum rgb, w = bits(n)
    float4* statSum2;       // sum of squares rgb, w = running bestZ
    unsigned long long* rayCounters;  // [0] shadow hops, [1] closest-hit rays of tail_kernel
};
// the slot a path was generated in (its pixel and sample index follow from it)
__device__ __forceinline__ uint32_t path_origin(const PathPool& pool, uint32_t slot) {
    return slot >= pool.baseCap ? pool.origin[slot - pool.baseCap] : slot;
}
