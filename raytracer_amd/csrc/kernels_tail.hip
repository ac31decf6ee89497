// tail_kernel: the rest of a pass for its last few paths, in one kernel (see the comment at the kernel). Its own
// translation unit because it holds a closest-hit traversal: like kernels_trace.hip it is built with IEEE division and
// square root, so a path's hits do not depend on whether its late bounces ran here or through extend_kernel.
#include "shade_body.h"

// The tail of a pass: once few paths are left (deep bounces of glass / mirror chains), a bounce costs the latency of
// its longest ray three times over (extend, shade, shadow launches plus a host round trip), whatever the count. Here
// FOUR lanes walk one path to its end without leaving the kernel. All four run the shading body on the same inputs
// (same results, same stores); what they share out are the traversals, which are independent of one another and are
// the latency that counts: lane 0 traces the next ray of the path while lanes 1..3 trace the shadow chains of the
// vertex just shaded (rounds of four jobs when the scene has more than three lights). Every job is a chain of
// closest hits through one loop - the extend job is a chain that ends after its first hop - so the four lanes stay in
// one instruction stream. Results go round the quad by shuffles; each lane then writes the transmittances and resolves
// the vertex (resolve_slot), so a lane only ever reads what it has written itself.
#ifndef TAIL_REFILL_MIN
#define TAIL_REFILL_MIN 2u   // idle quads of a wave before it fetches new paths
#endif
template <bool MICRO, bool XF>
#ifdef TAIL_WAVES_PER_EU
__attribute__((amdgpu_waves_per_eu(TAIL_WAVES_PER_EU)))
#endif
__global__ __launch_bounds__(TRAV_BLOCK) void tail_kernel(DevScene scArg, DevCamera cam, PathPool pool, DevPhotonMap pm, SegQ qIn,
                                                          const uint32_t* __restrict__ queueIn, int bounce0, int maxBounces,
                                                          unsigned long long seed, int sBase, int S, const int* __restrict__ pixels,
                                                          int usePhotons, uint2* spill, uint32_t chunk) {
    __shared__ uint2 lds_stack[STACK_LDS * TRAV_BLOCK];
    const DevSceneT<XF> sc(scArg);
    const uint32_t gtid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, q = lane & 3u;
    const int quadBase = (int)(lane & ~3u);
    TravStack st = make_trav_stack(&lds_stack[threadIdx.x], spill, gtid, gridDim.x * blockDim.x);
    const int nL = sc.n_lights;
    const uint32_t cap = pool.capacity, mult = (uint32_t)max(1, nL);
    uint32_t hops = 0, rays = 0;
    // persistent grid: the 16 quads of a wave take paths from the queue as they become free (segqueue.h)
    SegWalk walk;
    segq_walk_init(qIn, walk, gtid >> 6, chunk);
    bool exhausted = false;
    uint32_t slot = 0u, oslot = 0u;   // oslot: the slot the path was generated in (pathstate.h "Compaction"): its shadow entries are kept there
    bool needExtend = false;    // the path's next ray waits in pool.rays[slot]
    bool haveVertex = false;    // the NEE of the vertex shaded last waits for its shadow results
    int bounce = bounce0;       // bounce index of that next ray
    for (;;) {
        const bool busy = needExtend || haveVertex;   // the same in the four lanes of a quad
        const unsigned long long idleQuads = __ballot(!busy && q == 0u);
        const uint32_t nIdle = (uint32_t)__popcll(idleQuads);
        if (!exhausted && (nIdle >= TAIL_REFILL_MIN || __ballot(busy) == 0ull)) {
            uint32_t first = 0;
            const uint32_t got = segq_take(qIn, walk, nIdle, first);
            if (got == 0u) exhausted = true;
            const uint32_t rank = (uint32_t)__popcll(idleQuads & ((1ull << quadBase) - 1ull));   // idle quads below this one
            if (!busy && rank < got) {
                slot = queueIn[first + rank];
                oslot = path_origin(pool, slot);
                needExtend = true;
                haveVertex = false;
                bounce = bounce0;
            }
        }
        if (__ballot(needExtend || haveVertex) == 0ull) {
            if (exhausted) break;
            continue;
        }
        if (!(needExtend || haveVertex)) continue;
        // ---- one step of this quad's path: trace (next ray + shadows of the last vertex), resolve, shade ----
        RayT ray{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f};
        uint32_t rayFlags = 0u;   // the path's PF_ flags travel with its ray (shade_body.h)
        if (needExtend) {
            const float4 ra = pool.rays[2 * (size_t)slot], rb = pool.rays[2 * (size_t)slot + 1];
            ray = RayT{ra.x, ra.y, ra.z, rb.x, rb.y, rb.z, ra.w};
            rayFlags = __float_as_uint(rb.w);
        }
        HitT hit{RTX_BIG, 0.0f, 0.0f, -1};
        const int nJobs = 1 + (haveVertex ? nL : 0);
        const bool useTransmittance = (bounce - 1) <= 8;   // of the vertex whose shadows are traced (RayCaster.cpp:454)
        for (int r = 0; r * 4 < nJobs; ++r) {
            const int j = r * 4 + (int)q;
            const bool isExtend = j == 0;
            bool act = false;
            f3 o = mk3(0.0f), d = mk3(0.0f, 0.0f, 1.0f);
            float maxDist = 0.0f;
            if (isExtend) {
                if (needExtend) {
                    o = mk3(ray.ox, ray.oy, ray.oz);
                    d = mk3(ray.dx, ray.dy, ray.dz);
                    maxDist = ray.tmax;
                    act = true;
                }
            } else if (haveVertex && j - 1 < nL) {
                if (pool.nee[(size_t)(j - 1) * cap + slot].w >= 0.0f) {
                    const size_t e = (size_t)oslot * mult + (size_t)(j - 1);
                    const float4 a = pool.shadowQ[2 * e], b = pool.shadowQ[2 * e + 1];
                    o = mk3(a.x, a.y, a.z);
                    d = mk3(b.x, b.y, b.z);
                    maxDist = a.w;
                    act = true;
                }
            }
            // one chain of closest hits per lane
            float visibility = 1.0f, result = 1.0f;
            int hop = 0;
            HitT h{RTX_BIG, 0.0f, 0.0f, -1};
            bool running = act;
            while (running) {
                const RayT seg{o.x, o.y, o.z, d.x, d.y, d.z, maxDist};
                uint32_t nc = 0, tc = 0;
                h = traverse_closest<false>(sc, seg, RTX_HIT_FRONT_AND_BACK, st, &nc, &tc);
                intersect_lights(sc, seg, h);
                if (isExtend) {
                    ++rays;
                    running = false;
                } else {
                    ++hops;
                    running = !shadow_hop_end(sc, o, d, maxDist, h, useTransmittance, hop, visibility, result);
                }
            }
            // results round the quad
            if (r == 0) {
                hit.t = __shfl(h.t, quadBase);
                hit.u = __shfl(h.u, quadBase);
                hit.v = __shfl(h.v, quadBase);
                hit.slot = __shfl(h.slot, quadBase);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float Tk = __shfl(result, quadBase + k);
                const int ak = __shfl((int)act, quadBase + k);
                const int jk = r * 4 + k;
                if (jk >= 1 && ak) pool.nee[(size_t)(jk - 1) * cap + slot].w = Tk;
            }
        }
        if (haveVertex) resolve_slot(pool, slot, nL, usePhotons != 0);
        haveVertex = false;
        if (needExtend) {
            // MICRO here = "general": microfacet or Phong materials, textured specular terms, or a caustics map (launch_tail)
            needExtend = shade_body<MICRO, true, MICRO, MICRO>(sc, cam, pool, pm, slot, ray, rayFlags, hit, bounce, maxBounces, seed, sBase, S, pixels, SegQ{nullptr});
            // the NEE entries are this vertex's only when the body left something pending (a path that ended on a miss
            // or a light never reached its NEE loop: the entries then hold an older vertex's values)
            haveVertex = pool.pendBeta[slot].w >= 0.0f;
            ++bounce;
        }
    }
    atomicAdd(&pool.rayCounters[0], (unsigned long long)hops);
    atomicAdd(&pool.rayCounters[1], (unsigned long long)rays);
}

void launch_tail(const DevScene& sc, const LaunchCfg& cfg, const DevCamera& cam, const PathPool& pool, const DevPhotonMap& pm,
                 const SegQ& qIn, const uint32_t* queueIn, uint32_t n, int bounce0, int maxBounces, unsigned long long seed, int sBase,
                 int S, const int* pixels, int hasMicrofacet, int usePhotons) {
    if (!n) return;
    const uint32_t need = (4u * n + TRAV_BLOCK - 1) / TRAV_BLOCK;   // 4 lanes per path
    const uint32_t grid = need < (uint32_t)cfg.gridBlocks ? need : (uint32_t)cfg.gridBlocks;   // the spill slab covers gridBlocks blocks
    // paths reserved per atomic by a wave of 16 quads: small enough that every wave of the grid gets some
    uint32_t chunk = n / (grid * (TRAV_BLOCK / 64u) * 2u);
    chunk = chunk < 16u ? 16u : (chunk > 256u ? 256u : (chunk / 16u) * 16u);
#define RTX_LAUNCH_TAIL(M, X)                                                                                                         \
    hipLaunchKernelGGL((tail_kernel<M, X>), dim3(grid), dim3(TRAV_BLOCK), 0, cfg.stream, sc, cam, pool, pm, qIn, queueIn, bounce0, maxBounces, \
                       seed, sBase, S, pixels, usePhotons, cfg.spill, chunk)
    if (hasMicrofacet || usePhotons) {   // hasMicrofacet: any material beyond plain Blinn (rtx_render.cpp)
        if (sc.has_xf) RTX_LAUNCH_TAIL(true, true);
        else RTX_LAUNCH_TAIL(true, false);
    } else {
        if (sc.has_xf) RTX_LAUNCH_TAIL(false, true);
        else RTX_LAUNCH_TAIL(false, false);
    }
#undef RTX_LAUNCH_TAIL
}
