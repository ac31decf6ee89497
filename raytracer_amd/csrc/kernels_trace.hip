// HIP kernels for ray casting on gfx950: extend (closest hit), pixel-centre ray generation, hit expansion and
// the shadow-chain visibility query. See traverse.h / shadow.h for the device functions and their reference
// citations.
#define RTX_TRAV_FAST_RCP 1   // traverse.h trav_init: 1 / dir by v_rcp_f32 in this translation unit (extend_kernel, visibility_kernel)
#include "kernels.h"
#include "shadow.h"
#include "segqueue.h"

// ---- segmented queue control (segqueue.h) ---------------------------------------------------------------------
// One wave per queue: lane s owns segment s.
__global__ void segq_scan_kernel(SegQ qa, uint32_t Ga, uint32_t Ba, uint32_t multA, SegQ qb, uint32_t Gb, uint32_t Bb, uint32_t multB,
                                 uint32_t* totals, uint32_t slotA, uint32_t slotB, const uint32_t* blocksFrom) {
    const SegQ q = blockIdx.x == 0 ? qa : qb;
    uint32_t G = blockIdx.x == 0 ? Ga : Gb;
    const uint32_t B = blockIdx.x == 0 ? Ba : Bb, mult = blockIdx.x == 0 ? multA : multB;
    if (blocksFrom) G = max(1u, QSEG * segq_nee_rows(*blocksFrom, B));
    const uint32_t l = threadIdx.x;
    const uint32_t c = q.ctl[SEGQ_COUNT + l * QSTRIDE];
    uint32_t x = c;   // inclusive scan over the 64 segments
    for (uint32_t d = 1; d < 64u; d <<= 1) {
        const uint32_t y = __shfl_up(x, d);
        if (l >= d) x += y;
    }
    q.ctl[SEGQ_PREFIX + l] = x - c;
    if (l == 63u) {
        q.ctl[SEGQ_PREFIX + QSEG] = x;
        totals[blockIdx.x == 0 ? slotA : slotB] = x;
    }
    q.ctl[SEGQ_START + l] = segq_first_block(l, G, q.run) * B * mult;
    q.ctl[SEGQ_COUNT + l * QSTRIDE] = 0u;
    q.ctl[SEGQ_CURSOR + l * QSTRIDE] = 0u;
}
__global__ void segq_identity_kernel(SegQ q, uint32_t n) {
    const uint32_t l = threadIdx.x;
    const uint32_t lo = (uint32_t)(((unsigned long long)l * n) / QSEG);
    q.ctl[SEGQ_PREFIX + l] = lo;
    q.ctl[SEGQ_START + l] = lo;
    if (l == 63u) q.ctl[SEGQ_PREFIX + QSEG] = n;
    q.ctl[SEGQ_COUNT + l * QSTRIDE] = 0u;
    q.ctl[SEGQ_CURSOR + l * QSTRIDE] = 0u;
}
void launch_segq_scan(const SegQ& q, uint32_t producerBlocks, uint32_t blockThreads, uint32_t mult, uint32_t* totals,
                      uint32_t totalsSlot, hipStream_t stream, const uint32_t* blocksFrom) {
    hipLaunchKernelGGL(segq_scan_kernel, dim3(1), dim3(64), 0, stream, q, producerBlocks ? producerBlocks : 1u, blockThreads, mult, q, 1u,
                       blockThreads, mult, totals, totalsSlot, totalsSlot, blocksFrom);
}
void launch_segq_scan2(const SegQ& qa, uint32_t Ga, uint32_t multA, const SegQ& qb, uint32_t Gb, uint32_t multB, uint32_t blockThreads,
                       uint32_t* totals, hipStream_t stream) {
    hipLaunchKernelGGL(segq_scan_kernel, dim3(2), dim3(64), 0, stream, qa, Ga ? Ga : 1u, blockThreads, multA, qb, Gb ? Gb : 1u, blockThreads,
                       multB, totals, 0u, 1u, (const uint32_t*)nullptr);
}
void launch_segq_identity(const SegQ& q, uint32_t n, hipStream_t stream) {
    hipLaunchKernelGGL(segq_identity_kernel, dim3(1), dim3(64), 0, stream, q, n);
}

// Persistent extend kernel. Lanes pull rays from a segmented queue: a wave refills its idle lanes (one atomic for
// the group, slots handed out by a ballot prefix) whenever at least REFILL_MIN lanes are idle or nothing is left to
// do, so a wave keeps most lanes busy although rays finish at very different times. Each iteration runs ONE kind of
// step, inner node or leaf, for the lanes that are at that kind of node; the other lanes wait for their turn, so
// neither code path runs for a handful of lanes while the rest of the wave idles.
// Every wave leaves the loop once the queue is drained and all of its lanes have finished.
// A wave runs the leaf code when at least 1/LEAF_WEIGHT as many lanes wait at a leaf as want to descend.
#ifndef LEAF_WEIGHT
#define LEAF_WEIGHT 1u
#endif
#ifndef REFILL_MIN
#define REFILL_MIN 16
#endif
#ifndef EXTEND_WAVES_PER_EU
#define EXTEND_WAVES_PER_EU 6
#endif
// the general instantiation (XF: object-space triangles, Sphere / Plane, the reference's box tests) needs more registers:
// at six waves it spills 43 of them inside the loop - C3 scene through the general kernels, 32 spp: extend 211 / 153 / 125 /
// 125 ms at 6 / 5 / 4 / 3 waves per SIMD (the lean kernel: 82 ms)
#ifndef EXTEND_WAVES_PER_EU_GENERAL
#define EXTEND_WAVES_PER_EU_GENERAL 4
#endif
// BOTH: hitSide is RTX_HIT_FRONT_AND_BACK (every ray of the render loop): the side test of objects.cpp:175-191 - a normalised
// geometric normal per candidate - is then not compiled into the leaf step (19 ms of the C3 frame's extend time)
template <bool COUNT, bool XF, bool BOTH, bool FRONT = false>
__global__ __launch_bounds__(TRAV_BLOCK) __attribute__((amdgpu_waves_per_eu(XF ? EXTEND_WAVES_PER_EU_GENERAL : EXTEND_WAVES_PER_EU))) void extend_kernel(DevScene scArg, const float4* __restrict__ rays, SegQ q,
                                                            const uint32_t* __restrict__ items, int hitSide, int withLights,
                                                            float4* __restrict__ hits, uint2* spill, unsigned long long* counters,
                                                            uint32_t chunk) {
    __shared__ uint2 lds_stack[STACK_LDS * TRAV_BLOCK];
    const DevSceneT<XF> sc(scArg);
    const uint32_t gtid = blockIdx.x * blockDim.x + threadIdx.x;
    TravStack st = make_trav_stack<true>(&lds_stack[threadIdx.x], spill, gtid, gridDim.x * blockDim.x);
    uint32_t nodeCount = 0, triCount = 0, rayCount = 0;
    TravState s;
    // A lane's state is its s.cur alone, so that every predicate of the loop is one compare on a register (a loop-carried
    // bool is kept as a lane mask, and the compiler turns a ballot of it into a select and a compare per use):
    //   (int)cur >= 0   at an inner node          (int)cur < RTX_LANE_IDLE   at a leaf
    //   cur == RTX_TRAV_DONE (-1)  holds a finished ray, not yet retired      cur == RTX_LANE_IDLE (-2)  holds no ray
    s.cur = RTX_LANE_IDLE;
    s.sp = 0;
    bool exhausted = false;   // the queue is drained (wave-uniform)
    SegWalk walk;
    segq_walk_init<FRONT>(q, walk, gtid >> 6, chunk);
    uint32_t rayIndex = 0;
    for (;;) {
        // ---- retire finished rays in batches (a handful of lanes running the epilogue every iteration would cost
        //      more issue slots than the traversal steps themselves), then refill the idle lanes ----
        const bool fin = s.cur == RTX_TRAV_DONE;
        const unsigned long long finMask = __ballot(fin);
        const unsigned long long freeMask = finMask | __ballot(s.cur == RTX_LANE_IDLE);
        if (finMask != 0ull && ((uint32_t)__popcll(freeMask) >= REFILL_MIN || freeMask == ~0ull)) {
            if (fin && trav_verify(s, sc)) {   // (general kernels: a hit the reference's box tests cull is traced again without that object)
                if (withLights) intersect_lights(sc, s.r, s.best);
                hits[rayIndex] = make_float4(s.best.t, s.best.u, s.best.v, __int_as_float(s.best.slot));
                if (COUNT) rayCount++;
                s.cur = RTX_LANE_IDLE;
            }
        }
        const unsigned long long idle = __ballot(s.cur == RTX_LANE_IDLE);
        const uint32_t nIdle = (uint32_t)__popcll(idle);
        if (!exhausted && nIdle >= REFILL_MIN) {
            uint32_t first = 0;
            const uint32_t got = segq_take<FRONT>(q, walk, nIdle, first);
            if (got == 0u) exhausted = true;
            if (s.cur == RTX_LANE_IDLE) {
                const uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                if (prefix < got) {
                    rayIndex = items ? items[first + prefix] : first + prefix;   // path slots or identity
                    const float4 a = rays[2 * (size_t)rayIndex], b = rays[2 * (size_t)rayIndex + 1];
                    trav_init(s, sc, RayT{a.x, a.y, a.z, b.x, b.y, b.z, a.w});
                }
            }
        }
        // ---- one traversal step for the larger group of lanes ----
        {
            const bool atInner = (int)s.cur >= 0;
            const bool atLeaf = (int)s.cur < (int)RTX_LANE_IDLE;
            const unsigned long long innerMask = __ballot(atInner), leafMask = __ballot(atLeaf);
            if ((innerMask | leafMask) == 0ull) {
                // nothing to step: finished rays are retired at the top of the loop, a drained queue ends it
                if (exhausted && __ballot(s.cur == RTX_TRAV_DONE) == 0ull) break;
                continue;
            }
            const uint32_t nInner = (uint32_t)__popcll(innerMask), nLeaf = (uint32_t)__popcll(leafMask);
#ifdef TRAV_PROFILE
            if (COUNT && (threadIdx.x & 63u) == 0u) {   // wave-level step statistics (variant builds only)
                const bool leafStep = nLeaf * LEAF_WEIGHT >= nInner;
                atomicAdd(&counters[leafStep ? 5 : 3], 1ull);
                atomicAdd(&counters[leafStep ? 6 : 4], (unsigned long long)(leafStep ? nLeaf : nInner));
                atomicAdd(&counters[7], (unsigned long long)__popcll(__ballot(true)));
            }
#endif
            if (nLeaf * LEAF_WEIGHT >= nInner) {
                if (atLeaf) {
                    const uint32_t nt = trav_leaf_step(s, sc, st, BOTH ? RTX_HIT_FRONT_AND_BACK : hitSide);
                    if (COUNT) triCount += nt;
                }
            } else if (atInner) {
                if (COUNT) nodeCount++;
                trav_inner_step(s, sc, st);
            }
        }
    }
    if (COUNT) {
        atomicAdd(&counters[0], (unsigned long long)nodeCount);
        atomicAdd(&counters[1], (unsigned long long)triCount);
        atomicAdd(&counters[2], (unsigned long long)rayCount);
    }
}

void launch_extend_queue(const DevScene& sc, const LaunchCfg& cfg, const float4* rays, const SegQ& q, const uint32_t* items,
                         uint32_t n, int hitSide, int withLights, float4* hits) {
    if (n == 0) return;
    const uint32_t need = (n + TRAV_BLOCK - 1) / TRAV_BLOCK;
    const uint32_t grid = need < (uint32_t)cfg.gridBlocks ? need : (uint32_t)cfg.gridBlocks;
    const uint32_t chunk = segq_chunk(n, grid * (TRAV_BLOCK / 64));
#define RTX_LAUNCH_EXTEND(C, X, B)                                                                                                  \
    hipLaunchKernelGGL((extend_kernel<C, X, B>), dim3(grid), dim3(TRAV_BLOCK), 0, cfg.stream, sc, rays, q, items, hitSide, withLights, \
                       hits, cfg.spill, cfg.counters, chunk)
#define RTX_LAUNCH_EXTEND_FRONT(X, B)                                                                                                     \
    hipLaunchKernelGGL((extend_kernel<false, X, B, true>), dim3(grid), dim3(TRAV_BLOCK), 0, cfg.stream, sc, rays, q, items, hitSide, withLights, \
                       hits, cfg.spill, cfg.counters, chunk)
    if (!cfg.counting && segq_front(n)) {   // a launch of middling size walks its queue as one front (segqueue.h SegWalk)
        switch ((sc.has_xf ? 2 : 0) | (hitSide == RTX_HIT_FRONT_AND_BACK ? 1 : 0)) {
            case 0: RTX_LAUNCH_EXTEND_FRONT(false, false); break;
            case 1: RTX_LAUNCH_EXTEND_FRONT(false, true); break;
            case 2: RTX_LAUNCH_EXTEND_FRONT(true, false); break;
            default: RTX_LAUNCH_EXTEND_FRONT(true, true); break;
        }
        return;
    }
    switch ((cfg.counting ? 4 : 0) | (sc.has_xf ? 2 : 0) | (hitSide == RTX_HIT_FRONT_AND_BACK ? 1 : 0)) {
        case 0: RTX_LAUNCH_EXTEND(false, false, false); break;
        case 1: RTX_LAUNCH_EXTEND(false, false, true); break;
        case 2: RTX_LAUNCH_EXTEND(false, true, false); break;
        case 3: RTX_LAUNCH_EXTEND(false, true, true); break;
        case 4: RTX_LAUNCH_EXTEND(true, false, false); break;
        case 5: RTX_LAUNCH_EXTEND(true, false, true); break;
        case 6: RTX_LAUNCH_EXTEND(true, true, false); break;
        default: RTX_LAUNCH_EXTEND(true, true, true); break;
    }
#undef RTX_LAUNCH_EXTEND
#undef RTX_LAUNCH_EXTEND_FRONT
}
void launch_extend(const DevScene& sc, const LaunchCfg& cfg, const float4* rays, uint32_t n, int hitSide, int withLights,
                   float4* hits) {
    if (n == 0) return;
    launch_segq_identity(cfg.identQ, n, cfg.stream);
    launch_extend_queue(sc, cfg, rays, cfg.identQ, nullptr, n, hitSide, withLights, hits);
}

// CameraAdapter.h:7-27
__global__ void primary_raygen_kernel(DevCamera cam, float4* __restrict__ rays) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= cam.W || y >= cam.H) return;
    const float ndcX = (((float)x + 0.5f) / (float)cam.W) * 2.0f - 1.0f;
    const float ndcY = 1.0f - (((float)y + 0.5f) / (float)cam.H) * 2.0f;
    const float sx = ndcX * cam.halfX;
    const float sy = ndcY * cam.halfY;
    f3 dir = ld3(cam.r) * sx + ld3(cam.u) * sy + ld3(cam.f);
    dir = normalized3(dir);
    const size_t i = (size_t)y * cam.W + x;
    rays[2 * i] = make_float4(cam.pos[0], cam.pos[1], cam.pos[2], RTX_BIG);
    rays[2 * i + 1] = make_float4(dir.x, dir.y, dir.z, 0.0f);
}

void launch_primary_raygen(const DevCamera& cam, float4* rays, hipStream_t stream) {
    dim3 block(64, 4), grid((cam.W + 63) / 64, (cam.H + 3) / 4);
    hipLaunchKernelGGL(primary_raygen_kernel, grid, block, 0, stream, cam, rays);
}

__global__ void expand_hits_kernel(DevScene sc, const float4* __restrict__ rays, const float4* __restrict__ hits,
                                   uint32_t n, rtx_hit* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 a = rays[2 * (size_t)i], b = rays[2 * (size_t)i + 1];
    const float4 hv = hits[i];
    RayT r{a.x, a.y, a.z, b.x, b.y, b.z, a.w};
    HitT h{hv.x, hv.y, hv.z, __float_as_int(hv.w)};
    rtx_hit o;
    o.t = h.t;
    o.obj = -1;
    o.face = -1;
    o.mtl_id = 0;
    o.front = 1;
    o.light = -1;
    o.bu = 0.0f;
    o.bv = 0.0f;
    for (int k = 0; k < 3; ++k) {
        o.p[k] = o.N[k] = o.GN[k] = o.uvw[k] = o.dpdu[k] = o.dpdv[k] = 0.0f;
    }
    if (h.slot != -1) {
        SurfHit s;
        if (h.slot >= 0) resolve_triangle(sc, r, h, s);
        else resolve_light(sc.lights, r, h, s);
        o.obj = s.obj;
        o.face = s.face;
        o.mtl_id = s.mtlID;
        o.front = s.front ? 1 : 0;
        o.light = s.light;
        o.bu = h.u;
        o.bv = h.v;
        o.p[0] = s.p.x; o.p[1] = s.p.y; o.p[2] = s.p.z;
        o.N[0] = s.N.x; o.N[1] = s.N.y; o.N[2] = s.N.z;
        o.GN[0] = s.GN.x; o.GN[1] = s.GN.y; o.GN[2] = s.GN.z;
        o.uvw[0] = s.uvw.x; o.uvw[1] = s.uvw.y; o.uvw[2] = s.uvw.z;
        o.dpdu[0] = s.dpdu.x; o.dpdu[1] = s.dpdu.y; o.dpdu[2] = s.dpdu.z;
        o.dpdv[0] = s.dpdv.x; o.dpdv[1] = s.dpdv.y; o.dpdv[2] = s.dpdv.z;
    }
    out[i] = o;
}

void launch_expand_hits(const DevScene& sc, const float4* rays, const float4* hits, uint32_t n, rtx_hit* out,
                        hipStream_t stream) {
    if (n == 0) return;
    hipLaunchKernelGGL(expand_hits_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, sc, rays, hits, n, out);
}

template <bool XF>
__global__ __launch_bounds__(TRAV_BLOCK) void visibility_kernel(DevScene scArg, const float4* __restrict__ rays,
                                                                const float4* __restrict__ normals, uint32_t n,
                                                                int useTransmittance, float* __restrict__ vis,
                                                                uint32_t* workCounter, uint2* spill,
                                                                unsigned long long* hopCounter) {
    __shared__ uint2 lds_stack[STACK_LDS * TRAV_BLOCK];
    const DevSceneT<XF> sc(scArg);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t gtid = blockIdx.x * blockDim.x + threadIdx.x;
    TravStack st = make_trav_stack(&lds_stack[threadIdx.x], spill, gtid, gridDim.x * blockDim.x);
    uint32_t hops = 0, nc = 0, tc = 0;
    for (;;) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(workCounter, 64u);
        base = __builtin_amdgcn_readfirstlane(base);
        if (base >= n) break;
        const uint32_t i = base + lane;
        if (i < n) {
            const float4 a = rays[2 * (size_t)i], b = rays[2 * (size_t)i + 1];
            const float4 nn = normals[i];
            vis[i] = visible_chain<false>(sc, mk3(a.x, a.y, a.z), mk3(nn.x, nn.y, nn.z), mk3(b.x, b.y, b.z), a.w,
                                          useTransmittance != 0, st, &hops, &nc, &tc);
        }
    }
    if (hopCounter) atomicAdd(hopCounter, (unsigned long long)hops);
}

void launch_visibility(const DevScene& sc, const LaunchCfg& cfg, const float4* rays, const float4* normals, uint32_t n,
                       int useTransmittance, float* vis, unsigned long long* hopCounter) {
    if (n == 0) return;
    (void)hipMemsetAsync(cfg.workCounter, 0, sizeof(uint32_t), cfg.stream);
    const uint32_t need = (n + TRAV_BLOCK - 1) / TRAV_BLOCK;
    const uint32_t grid = need < (uint32_t)cfg.gridBlocks ? need : (uint32_t)cfg.gridBlocks;
    if (sc.has_xf)
        hipLaunchKernelGGL(visibility_kernel<true>, dim3(grid), dim3(TRAV_BLOCK), 0, cfg.stream, sc, rays, normals, n, useTransmittance, vis,
                           cfg.workCounter, cfg.spill, hopCounter);
    else
        hipLaunchKernelGGL(visibility_kernel<false>, dim3(grid), dim3(TRAV_BLOCK), 0, cfg.stream, sc, rays, normals, n, useTransmittance, vis,
                           cfg.workCounter, cfg.spill, hopCounter);
}

int query_resident_blocks(int which) {
    int nb = 0;
    hipError_t e;
    if (which == 0) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, extend_kernel<false, false, true>, TRAV_BLOCK, 0);
    else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, visibility_kernel<false>, TRAV_BLOCK, 0);
    if (e != hipSuccess || nb < 1) nb = 2;
    return nb;
}
