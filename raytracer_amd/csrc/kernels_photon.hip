// HIP kernels for photon mapping on gfx950: emission + tracing (PhotonMapBuilder::EmitPhotons / TracePhoton,
// reference RayTracer/PhotonMapBuilder.cpp:80-268), hash-grid construction and the irradiance estimate for host
// probes. One lane follows one photon through its <= maxBounces+1 segments; lanes pull photon ids 64 at a time.
#include "kernels.h"
#include "photon.h"
#include "photon_cell.h"
#include "bsdf.h"
#include "shadow.h"
#include "photon_gather.h"

template <bool XF>
#ifdef PHOTON_WAVES_PER_EU
__attribute__((amdgpu_waves_per_eu(PHOTON_WAVES_PER_EU)))
#endif
__global__ __launch_bounds__(TRAV_BLOCK) void photon_emit_kernel(DevScene scArg, int numPhotons, int first, int stride, int maxBounces,
                                                                 unsigned long long seed, DevPhotonRec* __restrict__ globalOut,
                                                                 uint32_t globalCap, DevPhotonRec* __restrict__ causticOut,
                                                                 uint32_t causticCap, uint32_t* counts, uint2* spill) {
    __shared__ uint2 lds_stack[STACK_LDS * TRAV_BLOCK];
    const DevSceneT<XF> sc(scArg);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t gtid = blockIdx.x * blockDim.x + threadIdx.x;
    TravStack st = make_trav_stack(&lds_stack[threadIdx.x], spill, gtid, gridDim.x * blockDim.x);
    // light selection table (PhotonMapBuilder.cpp:82-95)
    float totalIntensity = 0.0f;
    for (int j = 0; j < sc.n_lights; ++j) totalIntensity += sc.lights[j].intensity[0] + sc.lights[j].intensity[1] + sc.lights[j].intensity[2];
    const uint32_t nWork = (numPhotons > first) ? (uint32_t)((numPhotons - first + stride - 1) / stride) : 0u;
    uint32_t segments = 0, nc = 0, tc = 0;
    for (;;) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&counts[2], 64u);
        base = __builtin_amdgcn_readfirstlane(base);
        if (base >= nWork) break;
        const uint32_t w = base + lane;
        if (w >= nWork || totalIntensity <= 0.0f) continue;
        const uint32_t photonId = (uint32_t)first + w * (uint32_t)stride;
        const unsigned long long key = rng_key(seed, photonId, 0xFFFFFFFFu);
        // :113-134 light choice proportional to r+g+b, RandomPhoton, power / selection pdf
        const float rsel = rng_draw(key, 0, 0) * totalIntensity;
        float sum = 0.0f;
        int lightIndex = 0;
        for (int j = 0; j < sc.n_lights; ++j) {
            sum += sc.lights[j].intensity[0] + sc.lights[j].intensity[1] + sc.lights[j].intensity[2];
            if (rsel <= sum) {
                lightIndex = j;
                break;
            }
        }
        const DevLight& light = sc.lights[lightIndex];
        float u[4];
        for (int k = 0; k < 4; ++k) u[k] = rng_draw(key, 0, 1 + k);
        f3 ro, rd, power;
        light_random_photon(light, u, ro, rd, power);
        const float li = light.intensity[0] + light.intensity[1] + light.intensity[2];
        const float lightSelectionPdf = li / totalIntensity;
        power = power * (1.0f / lightSelectionPdf);

        bool causticPath = false, hasHitDiffuse = false, hasTransmitted = false;
        for (int bounce = 0; bounce <= maxBounces; ++bounce) {   // TracePhoton :149-268
            if (max3c(power) < 1e-6f) break;
            RayT r{ro.x, ro.y, ro.z, rd.x, rd.y, rd.z, RTX_BIG};
            ++segments;
            HitT h = traverse_closest<false>(sc, r, RTX_HIT_FRONT_AND_BACK, st, &nc, &tc);
            if (h.slot < 0) break;
            SurfHit sh;
            resolve_triangle(sc, r, h, sh);
            const int mi = sc.objs[sh.obj].material;
            if (mi < 0) break;
            const DevMaterial& mtl = sc.mtls[mi];
            apply_bump(sc, mtl, sh);
            const f3 hitPos = sh.p;
            const f3 N = normalized3(sh.N);
            const bool inside = dot3(rd, N) > 0.0f;
            const bool isDiffuseSurface = mtl_is_photon_surface(mtl);   // IsPhotonSurface
            const bool storeAsCaustic = causticPath && hasTransmitted && !hasHitDiffuse && isDiffuseSurface;
            if (bounce >= 1 && isDiffuseSurface) {
                const f3 d = normalized3(rd);
                DevPhotonRec rec;
                rec.pos = make_float4(hitPos.x, hitPos.y, hitPos.z, __uint_as_float(photonId * 4u + (uint32_t)bounce));
                rec.dir = make_float4(d.x, d.y, d.z, 0.0f);
                rec.power = make_float4(power.x, power.y, power.z, 0.0f);
                if (storeAsCaustic) {
                    const uint32_t k = atomicAdd(&counts[1], 1u);
                    if (k < causticCap) causticOut[k] = rec;
                } else {
                    const uint32_t k = atomicAdd(&counts[0], 1u);
                    if (k < globalCap) globalOut[k] = rec;
                }
                hasHitDiffuse = true;
                causticPath = false;
            }
            SampCtx sx;
            sx.P = sh.p;
            sx.V = -normalized3(rd);
            sx.N = sh.N;
            sx.GN = sh.GN;
            sx.uvw = sh.uvw;
            sx.front = !inside;
            sx.key = key;
            sx.stream = (uint32_t)(bounce + 1);
            f3 nextDir;
            DirInfo si;
            if (mtl.type == RTX_MTL_MICROFACET) {
                const MicroCtx mctx = build_micro_ctx(sc, mtl, sx);
                if (!sample_microfacet(mctx, sx, nextDir, si)) break;
            } else {
                const PhongCtx ctx = build_phong_ctx(sc, mtl, sx);
                if (!sample_phong_blinn(ctx, mtl.type == RTX_MTL_PHONG, sx, nextDir, si)) break;
            }
            if (!hasHitDiffuse && si.lobe == LOBE_SPECULAR && max3c(si.mult) > 0.0f) break;   // :225-227
            const float continueProbability = fminf(0.5f, max3c(si.mult));                   // :231-237
            if (continueProbability < 1e-6f) break;
            if (sx.rnd(DIM_RR) > continueProbability) break;
            const f3 adjustedPower = power * (1.0f / continueProbability);
            f3 nextPower = adjustedPower * si.mult;
            if (si.prob > 1e-6f) nextPower = nextPower * (1.0f / si.prob);
            else break;
            if (si.lobe == LOBE_DIFFUSE) causticPath = false;
            else if (si.lobe == LOBE_TRANSMISSION && !hasHitDiffuse) {
                causticPath = true;
                hasTransmitted = true;
            }
            const f3 Nn = inside ? -N : N;
            const f3 biasDir = normalized3(nextDir);
            ro = hitPos + (dot3(Nn, biasDir) >= 0.0f ? Nn : -Nn) * 1e-4f;
            rd = biasDir;
            power = nextPower;
        }
    }
    atomicAdd(&counts[3], segments);
}

// Bounding box of the stored photons as order-preserving integers (float bits, sign folded): out = {min xyz, max xyz}
__device__ __forceinline__ uint32_t float_order(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__global__ void photon_bbox_kernel(PhotonSet s, uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= s.n) return;
    const float4 p = s.raw[i].pos;
    atomicMin(&out[0], float_order(p.x)); atomicMin(&out[1], float_order(p.y)); atomicMin(&out[2], float_order(p.z));
    atomicMax(&out[3], float_order(p.x)); atomicMax(&out[4], float_order(p.y)); atomicMax(&out[5], float_order(p.z));
}
__global__ void photon_count_kernel(PhotonSet s, uint32_t* __restrict__ cellCount) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= s.n) return;
    atomicAdd(&cellCount[cell_of(s, s.raw[i].pos)], 1u);
}
__global__ void photon_probe_kernel(DevPhotonMap pm, const float* __restrict__ pos, const float* __restrict__ nrm, uint32_t n,
                                    float* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const f3 r = estimate_irradiance(pm, ld3(pos + 3 * (size_t)i), ld3(nrm + 3 * (size_t)i));
    out[3 * (size_t)i] = r.x;
    out[3 * (size_t)i + 1] = r.y;
    out[3 * (size_t)i + 2] = r.z;
}

void launch_photon_emit(const DevScene& sc, const LaunchCfg& cfg, int numPhotons, int first, int stride, int maxBounces,
                        unsigned long long seed, DevPhotonRec* g, uint32_t gcap, DevPhotonRec* c, uint32_t ccap, uint32_t* counts) {
    const uint32_t nWork = (numPhotons > first) ? (uint32_t)((numPhotons - first + stride - 1) / stride) : 0u;
    if (!nWork) return;
    const uint32_t need = (nWork + TRAV_BLOCK - 1) / TRAV_BLOCK;
    const uint32_t grid = need < (uint32_t)cfg.gridBlocks ? need : (uint32_t)cfg.gridBlocks;
    if (sc.has_xf)
        hipLaunchKernelGGL(photon_emit_kernel<true>, dim3(grid), dim3(TRAV_BLOCK), 0, cfg.stream, sc, numPhotons, first, stride, maxBounces,
                           seed, g, gcap, c, ccap, counts, cfg.spill);
    else
        hipLaunchKernelGGL(photon_emit_kernel<false>, dim3(grid), dim3(TRAV_BLOCK), 0, cfg.stream, sc, numPhotons, first, stride, maxBounces,
                           seed, g, gcap, c, ccap, counts, cfg.spill);
}
void launch_photon_bbox(const PhotonSet& s, uint32_t* out6, hipStream_t st) {
    if (!s.n) return;
    hipLaunchKernelGGL(photon_bbox_kernel, dim3((s.n + 255) / 256), dim3(256), 0, st, s, out6);
}
void launch_photon_count(const PhotonSet& s, uint32_t* cellCount, hipStream_t st) {
    if (!s.n) return;
    hipLaunchKernelGGL(photon_count_kernel, dim3((s.n + 255) / 256), dim3(256), 0, st, s, cellCount);
}
void launch_photon_probe(const DevPhotonMap& pm, const float* pos, const float* nrm, uint32_t n, float* out, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(photon_probe_kernel, dim3((n + 63) / 64), dim3(64), 0, st, pm, pos, nrm, n, out);
}
