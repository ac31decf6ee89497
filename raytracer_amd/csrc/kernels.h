// Host-callable launch wrappers of the HIP kernels (definitions in kernels_trace.hip / kernels_path.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "layout.h"
#include "segqueue.h"
#include "../../include/rtx.h"

// Camera with the per-frame constants of RayCaster.cpp:717-724 / CameraAdapter.h:8-21 evaluated once on the host.
struct DevCamera {
    float pos[3], f[3], r[3], u[3];
    float halfX, halfY;
    float focus, aperture;
    int W, H;
    int thinLens;       // camera->dof > 1e-6f
    int srgb;
    float fov;
};

struct LaunchCfg {
    int gridBlocks;          // persistent grid: CUs x resident blocks
    uint2* spill;            // gridBlocks*256*STACK_GLOBAL entries
    uint32_t* workCounter;   // zeroed before every launch that fetches work dynamically
    SegQ identQ;             // control words of the identity queue used by plain ray batches
    unsigned long long* counters;  // [0] nodes, [1] tris, [2] rays (only with counting)
    hipStream_t stream;
    int counting;
};

// extend: closest hit for n rays (rtx_ray layout, 2 x float4) -> hits float4 {t,u,v,bits(slot)}
void launch_extend(const DevScene& sc, const LaunchCfg& cfg, const float4* rays, uint32_t n, int hitSide, int withLights,
                   float4* hits);
// the same for the n rays whose indices are the entries of a segmented queue (items == nullptr: identity)
void launch_extend_queue(const DevScene& sc, const LaunchCfg& cfg, const float4* rays, const SegQ& q, const uint32_t* items,
                         uint32_t n, int hitSide, int withLights, float4* hits);
void launch_segq_scan2(const SegQ& qa, uint32_t Ga, uint32_t multA, const SegQ& qb, uint32_t Gb, uint32_t multB, uint32_t blockThreads,
                       uint32_t* totals, hipStream_t stream);
// CameraAdapter.h:7-27 pixel-centre rays for the whole image
void launch_primary_raygen(const DevCamera& cam, float4* rays, hipStream_t stream);
// compact hit -> full rtx_hit (device array)
void launch_expand_hits(const DevScene& sc, const float4* rays, const float4* hits, uint32_t n, rtx_hit* out,
                        hipStream_t stream);
// Visible()/VisibleSimple(): rays = segments (o, unit d, tmax = distance), normals float4 per ray
void launch_visibility(const DevScene& sc, const LaunchCfg& cfg, const float4* rays, const float4* normals, uint32_t n,
                       int useTransmittance, float* vis, unsigned long long* hopCounter);

// 3x3 median of a linear RGB image (device pointers; linear / rgb8 outputs optional)
void launch_median3(const float* src, int W, int H, int srgb, float* linear, unsigned char* rgb8, hipStream_t stream);

int query_resident_blocks(int which);  // blocks per CU for kernel 'which' (0 extend, 1 shadow)

// ---- wavefront path tracing (kernels_path.hip) ----
struct PathPool;
struct DevPhotonMap;
// raygen and shade return their grid size: the geometry segq_scan needs for the queues they append to
// initGuides: raygen also clears the first-hit guides and the pending-resolve marker (legacy mode; the path loop's first shade writes them)
uint32_t launch_raygen(const DevCamera& cam, const PathPool& pool, const int* pixels, uint32_t nPix, int S, int sBase, int indexStride,
                       const SegQ& qOut, uint32_t* queueOut, int initGuides, hipStream_t stream);
#ifndef RTX_SHADE_BLOCK
#define RTX_SHADE_BLOCK 256   // threads per block of shade_kernel (its queue regions are laid out by it: rtx_render.cpp)
#endif
uint32_t launch_shade(const DevScene& sc, const DevCamera& cam, const PathPool& pool, const DevPhotonMap& pm, const SegQ& qIn,
                      const uint32_t* queueIn, uint32_t n, int bounce, int maxBounces, unsigned long long seed, int sBase, int S,
                      const int* pixels, const SegQ& qOut, uint32_t* queueOut, const SegQ& qShadow, int materialKinds,
                      hipStream_t stream, uint32_t repackBase = 0u);   // materialKinds: bit 0 MtlMicrofacet, bit 1 MtlPhong present
// repackBase != 0: the bounce moves its paths to consecutive slots from repackBase on (pathstate.h "Compaction"); launch_unpack
// returns the results of the nUsed slots of the compaction area to the slots of the pass when the pass ends
void launch_unpack(const PathPool& pool, uint32_t nUsed, int nLights, int usePhotons, hipStream_t stream);
// the (vertex, light) pairs that shade launch (n vertices) has queued in pool.qN, one lane per pair: lobe evaluation, MIS
// weight, shadow segments appended to qShadow (kernels_nee.hip). The queue must have been scanned; returns the grid size.
uint32_t launch_nee(const DevScene& sc, const PathPool& pool, uint32_t n, const SegQ& qShadow, int materialKinds, hipStream_t stream, uint32_t pairs = 0u);
// the rest of a pass for its last n paths in one kernel (n <= RTX_TAIL_MAX_PATHS)
#ifndef RTX_TAIL_MAX_PATHS
#define RTX_TAIL_MAX_PATHS 65536u
#endif
void launch_tail(const DevScene& sc, const LaunchCfg& cfg, const DevCamera& cam, const PathPool& pool, const DevPhotonMap& pm,
                 const SegQ& qIn, const uint32_t* queueIn, uint32_t n, int bounce0, int maxBounces, unsigned long long seed, int sBase,
                 int S, const int* pixels, int hasMicrofacet, int usePhotons);
void launch_shadow(const DevScene& sc, const LaunchCfg& cfg, const PathPool& pool, const SegQ& q, uint32_t n, int useTransmittance);
// the chains a lean shadow launch set aside (PathPool::cullShadows), through the general kernel; qList: control words of its own
void launch_shadow_culled(const DevScene& scGeneral, const LaunchCfg& cfg, const PathPool& pool, const SegQ& qList, int useTransmittance);
void launch_film(const PathPool& pool, const rtx_film& film, const int* pixels, uint32_t nPix, int S, int W, int H, int filterType,
                 float filterRadius, int resetZ, float4* samplesOut, int sBase, int sppTotal, float radianceClamp, int nLights,
                 int usePhotons, hipStream_t stream);
// legacy shading mode (kernels_legacy.hip): TraceLegacy for the n camera samples whose primary hits are in pool.hits
void launch_legacy(const DevScene& sc, const LaunchCfg& cfg, const DevCamera& cam, const PathPool& pool, const DevPhotonMap& pmGlobal,
                   const DevPhotonMap& pmCaustic, const SegQ& qIn, const uint32_t* queueIn, uint32_t n, unsigned long long seed,
                   int sBase, int S, const int* pixels, int blinnOnly, void* state);
// bytes of `state`: the calls and frames of every lane of the persistent grid (allocated once per path pool)
size_t legacy_state_bytes(const LaunchCfg& cfg);
void launch_legacy_secondary(const DevScene& sc, const LaunchCfg& cfg, const DevPhotonMap& pmGlobal, const DevPhotonMap& pmCaustic,
                             const rtx_shade_point& at, const float* dirs, uint32_t n, int recursive, int child, unsigned long long seed,
                             float4* out);
void launch_adaptive(const PathPool& pool, const int* pixels, uint32_t nPix, float maxHalf, int sppMax, int* pixelsNext,
                     uint32_t* counter, hipStream_t stream);
void launch_fill_int(int* p, uint32_t n, int v, hipStream_t stream);
void launch_film_clear(const rtx_film& film, uint32_t n, hipStream_t stream);
void launch_stats_clear(const PathPool& pool, uint32_t n, hipStream_t stream);
void launch_film_resolve(const rtx_film& film, uint32_t n, int srgb, float* linear, unsigned char* rgb8, hipStream_t stream);
// dst += srcs[0 .. nSrc) (accum, count, guides by sum; z by min); all buffers on the device of `stream`; nSrc < RTX_GROUP_MAX
void launch_film_merge(const rtx_film& dst, const rtx_film* srcs, int nSrc, uint32_t n, hipStream_t stream);
void launch_stats_take(float* dstSum, float* dstSum2, const float* srcSum, const float* srcSum2, const int* pixels, uint32_t n,
                       hipStream_t stream);
// rtx_math.h routine `fn` on device arrays (rtx_debug_math)
void launch_math_probe(int fn, const float* x, const float* y, uint32_t n, float* out, hipStream_t stream);
