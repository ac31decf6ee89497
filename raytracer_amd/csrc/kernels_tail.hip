// tail_kernel: the rest of a pass for its last few paths, in one kernel (see the comment at the kernel). Its own
// translation unit because it holds a closest-hit traversal: like kernels_trace.hip it is built with IEEE division and
// square root, so a path's hits do not depend on whether its late bounces ran here or through extend_kernel.
#include "shade_body.h"

// The tail of a pass: once few paths are left (deep bounces of glass / mirror chains), a bounce costs the latency of
// its longest ray three times over (extend, shade, shadow launches plus a host round trip), whatever the count. Here
// one lane walks one path to its end - closest hit, shading, NEE visibility and its resolve - without leaving the
// kernel, so the remaining bounces cost one path's own latency instead of a launch sequence each.
template <bool MICRO>
__global__ __launch_bounds__(TRAV_BLOCK) void tail_kernel(DevScene sc, DevCamera cam, PathPool pool, DevPhotonMap pm, SegQ qIn,
                                                          const uint32_t* __restrict__ queueIn, uint32_t n, int bounce0, int maxBounces,
                                                          unsigned long long seed, int sBase, int S, const int* __restrict__ pixels,
                                                          uint2* spill) {
    __shared__ uint32_t qtab[SEGQ_TABLE_WORDS];
    __shared__ uint2 lds_stack[STACK_LDS * TRAV_BLOCK];
    segq_load_table(qIn, qtab);
    __syncthreads();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    TravStack st{&lds_stack[threadIdx.x], spill + i, gridDim.x * blockDim.x};
    uint32_t hops = 0, rays = 0;
    if (i < n) {
        const uint32_t slot = queueIn[segq_item_index(qtab, i)];
        for (int bounce = bounce0; bounce < maxBounces; ++bounce) {
            const float4 ra = pool.rays[2 * (size_t)slot], rb = pool.rays[2 * (size_t)slot + 1];
            const RayT ray{ra.x, ra.y, ra.z, rb.x, rb.y, rb.z, ra.w};
            uint32_t nc = 0, tc = 0;
            HitT h = traverse_closest<false>(sc, ray, RTX_HIT_FRONT_AND_BACK, st, &nc, &tc);
            intersect_lights(sc, ray, h);
            ++rays;
            if (!shade_body<MICRO, true>(sc, cam, pool, pm, slot, ray, h, bounce, maxBounces, seed, sBase, S, pixels, SegQ{nullptr}, st, hops))
                break;
        }
    }
    atomicAdd(&pool.rayCounters[0], (unsigned long long)hops);
    atomicAdd(&pool.rayCounters[1], (unsigned long long)rays);
}

void launch_tail(const DevScene& sc, const LaunchCfg& cfg, const DevCamera& cam, const PathPool& pool, const DevPhotonMap& pm,
                 const SegQ& qIn, const uint32_t* queueIn, uint32_t n, int bounce0, int maxBounces, unsigned long long seed, int sBase,
                 int S, const int* pixels, int hasMicrofacet) {
    if (!n) return;
    const uint32_t grid = (n + TRAV_BLOCK - 1) / TRAV_BLOCK;   // n <= RTX_TAIL_MAX_PATHS <= gridBlocks * TRAV_BLOCK: inside the spill slab
    if (hasMicrofacet)
        hipLaunchKernelGGL(tail_kernel<true>, dim3(grid), dim3(TRAV_BLOCK), 0, cfg.stream, sc, cam, pool, pm, qIn, queueIn, n, bounce0,
                           maxBounces, seed, sBase, S, pixels, cfg.spill);
    else
        hipLaunchKernelGGL(tail_kernel<false>, dim3(grid), dim3(TRAV_BLOCK), 0, cfg.stream, sc, cam, pool, pm, qIn, queueIn, n, bounce0,
                           maxBounces, seed, sBase, S, pixels, cfg.spill);
}
