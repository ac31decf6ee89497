// Wavefront path tracing on gfx950: ray generation, shade, shadow, NEE resolve, film accumulation and adaptive
// refinement. Together with the extend kernel (kernels_trace.hip) these replace RayCaster::RenderBucket +
// TracePath (reference RayTracer/RayCaster.cpp:339-574, 697-1023). Kernels are connected by index queues in HBM
// that are compacted with wave64 ballot / mbcnt prefix sums (one atomic per wave).
#include "kernels.h"
#include "pathstate.h"
#include "bsdf.h"
#include "shadow.h"
#include "photon_gather.h"
#include "segqueue.h"
#include "shade_body.h"

// ---- wave-level queue append -------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_append(bool pred, uint32_t* counter) {
    const unsigned long long mask = __ballot(pred);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t total = (uint32_t)__popcll(mask);
    uint32_t base = 0;
    if (total) {
        const int leader = __ffsll((long long)mask) - 1;
        if ((int)lane == leader) base = atomicAdd(counter, total);
        base = __shfl(base, leader);
    }
    const uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    return base + prefix;
}

// ---- sampler (RayCaster.cpp:65-81, 784-796, 729-762) --------------------------------------------------------
// The reference's loop runs while digits are left (`for (i = index; i > 0; i /= BASE)`) and divides f by the base once per digit.
// Here it runs over every digit a positive int can have: the digits beyond the first add f * 0 = +0 to r, which leaves r as it
// is, and with the trip count fixed the loop unrolls and the chain f = 1/B, f/B, f/B/B, ... is folded at compile time (one IEEE
// division per step there as here), so no division is left in the kernel. (raygen_kernel is 1-2 ms of a C3 frame either way.)
template <int BASE>
__device__ __forceinline__ float halton(int index) {
    constexpr int DIGITS = BASE == 2 ? 31 : BASE == 3 ? 20 : BASE == 5 ? 14 : BASE == 7 ? 12 : 31;
    static_assert(BASE == 2 || BASE == 3 || BASE == 5 || BASE == 7, "digit counts above are for these bases");
    float r = 0.0f;
    if (index <= 0) return r;
    float f = 1.0f / (float)BASE;
    uint32_t i = (uint32_t)index;
#pragma unroll
    for (int k = 0; k < DIGITS; ++k) {
        r += f * (float)(i % (uint32_t)BASE);
        i /= (uint32_t)BASE;
        f /= (float)BASE;
    }
    return r;
}
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long& x) {
    unsigned long long z = (x += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
__device__ __forceinline__ void pixel_shifts(int x, int y, float& sx, float& sy) {
    unsigned long long s = ((unsigned long long)(uint32_t)x << 32) | (unsigned long long)(uint32_t)y;
    const unsigned long long h1 = splitmix64(s);
    const unsigned long long h2 = splitmix64(s);
    sx = (float)((double)(h1 & 0xFFFFFFFFULL) / 4294967296.0);
    sy = (float)((double)(h2 & 0xFFFFFFFFULL) / 4294967296.0);
}

// initGuides: also clear the first-hit guides and the pending-resolve marker of every slot. The path loop does not
// need it (its first shade writes all three for every path); the legacy kernel does.
__global__ void raygen_kernel(DevCamera cam, PathPool pool, const int* __restrict__ pixels, uint32_t nPix, int S, int sBase,
                              int indexStride, SegQ qOut, uint32_t* __restrict__ queueOut, int initGuides) {
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t nSlots = nPix * (uint32_t)S;
    bool valid = false;
    if (slot < nSlots) {
        const uint32_t lane = slot & 63u;
        const uint32_t cs = slot >> 6;
        const uint32_t chunk = cs / (uint32_t)S, sIn = cs % (uint32_t)S;
        const int pix = pixels[chunk * 64u + lane];
        if (pix >= 0) {
            valid = true;
            const int x = pix % cam.W, y = pix / cam.W;
            const int s = sBase + (int)sIn;
            float shiftX, shiftY;
            pixel_shifts(x, y, shiftX, shiftY);
            const int sampleIndex = pix * indexStride + (s + 1);
            const float hx = halton<2>(sampleIndex);
            const float hy = halton<3>(sampleIndex);
            float jx = hx + shiftX;
            float jy = hy + shiftY;
            jx = fmodf(jx, 1.0f) - 0.5f;
            jy = fmodf(jy, 1.0f) - 0.5f;
            const float samplePosX = (float)x + 0.5f + jx;
            const float samplePosY = (float)y + 0.5f + jy;
            const float ndcY = 1.0f - (samplePosY / (float)cam.H) * 2.0f;
            const float sy = ndcY * cam.halfY;
            const float ndcX = (samplePosX / (float)cam.W) * 2.0f - 1.0f;
            const float sx = ndcX * cam.halfX;
            const f3 r = ld3(cam.r), u = ld3(cam.u), f = ld3(cam.f), pos = ld3(cam.pos);
            const f3 dirVec = r * sx + u * sy + f;
            f3 ro, rd;
            if (cam.thinLens) {
                const f3 pixelWorld = pos + dirVec * cam.focus;
                const float lx = halton<5>(sampleIndex);
                const float ly = halton<7>(sampleIndex);
                const float rdisk = sqrtf(lx);
                const float theta = 2.0f * PI_F * ly;
                const float dx = rdisk * rtx_cosf(theta);
                const float dy = rdisk * rtx_sinf(theta);
                const f3 lensPoint = pos + r * (dx * cam.aperture) + u * (dy * cam.aperture);
                ro = lensPoint;
                rd = normalized3(pixelWorld - lensPoint);
            } else {
                ro = pos;
                rd = normalized3(dirVec);
            }
            pool.rays[2 * (size_t)slot] = make_float4(ro.x, ro.y, ro.z, RTX_BIG);
            pool.rays[2 * (size_t)slot + 1] = make_float4(rd.x, rd.y, rd.z, 0.0f);
            // L / beta / flags start as {0, BIG}, {1, 1}, PF_LAST_DELTA: the shading of bounce 0 assumes that (shade_body.h)
            if (initGuides) {
                pool.pendBeta[slot] = make_float4(0.0f, 0.0f, 0.0f, -1.0f);   // nothing to resolve (resolve_add)
                pool.albedo[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                pool.normal[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
            pool.samplePos[slot] = make_float2(samplePosX, samplePosY);
        }
    }
    __shared__ uint32_t qtmp[SEGQ_BLOCK_WORDS];
    const uint32_t q = segq_append_block(qOut, valid, 1u, gridDim.x, qtmp);
    if (valid) queueOut[q] = slot;
}

// ---- shade (per-vertex body: shade_body.h) -----------------------------------------------------------------------
// Waves per SIMD: the instantiation for plain Blinn scenes without a caustics map fits 128 registers without a spill and runs
// four (C3 frame: 767 ms against 776 at three); the ones with the microfacet / Phong forms or the photon gather compiled in
// would spill 16-24 registers there and stay at three.
#ifndef SHADE_WAVES_PER_EU
#define SHADE_WAVES_PER_EU 4
#endif
#ifndef SHADE_WAVES_PER_EU_GENERAL
#define SHADE_WAVES_PER_EU_GENERAL 3
#endif
// (the instantiations with the microfacet / Phong forms compiled in spill 36-74 registers at three waves, yet two waves
// are slower: C3 scene through them, 32 spp: 96-105 ms against 110-112; the plain form: 91)
// LDSTAB: the scene's object, material and texture-map tables are copied to LDS when the block starts (they are small: 22 + 17 + 3
// records for cerealBar.xml) and the shading body reads them there. A vertex reaches its texels through hit -> triangle ->
// object -> material -> texture map -> texel, each step a memory round trip the wave waits out; the kernel waits on memory for
// two thirds of its time, and three of those six steps are lookups in these tables.
#define SHADE_LDS_OBJS 64
#define SHADE_LDS_MTLS 32
#define SHADE_LDS_TEX 8
template <class T>
__device__ __forceinline__ void shade_copy_table(T* dst, const T* src, int count) {
    const uint32_t words = (uint32_t)count * (uint32_t)(sizeof(T) / 4u);
    for (uint32_t k = threadIdx.x; k < words; k += blockDim.x) reinterpret_cast<uint32_t*>(dst)[k] = reinterpret_cast<const uint32_t*>(src)[k];
}
// REPACK: this bounce moves the paths together (pathstate.h "Compaction"): the vertex at queue position i is read at its slot and
// written to slot wbase + i of the compaction area, where the path lives from here on.
template <bool MICRO, bool PHOT, bool PHONG, bool LDSTAB, bool REPACK = false>
__global__ __launch_bounds__(RTX_SHADE_BLOCK) __attribute__((amdgpu_waves_per_eu((MICRO || PHOT || PHONG) ? SHADE_WAVES_PER_EU_GENERAL : SHADE_WAVES_PER_EU))) void shade_kernel(DevScene scArg, DevCamera cam, PathPool pool, DevPhotonMap pm,
                                                    SegQ qIn, const uint32_t* __restrict__ queueIn, uint32_t n, int bounce,
                                                    int maxBounces, unsigned long long seed, int sBase, int S,
                                                    const int* __restrict__ pixels, SegQ qOut, uint32_t* __restrict__ queueOut,
                                                    SegQ qShadow, SegQ qNee, uint32_t wbase) {
    __shared__ uint32_t qtab[SEGQ_TABLE_WORDS];
    __shared__ ObjRec sObjs[LDSTAB ? SHADE_LDS_OBJS : 1];
    __shared__ DevMaterial sMtls[LDSTAB ? SHADE_LDS_MTLS : 1];
    __shared__ DevTexMap sTex[LDSTAB ? SHADE_LDS_TEX : 1];
    DevScene sc = scArg;
    if (LDSTAB) {   // (the launch wrapper has checked that the tables fit)
        shade_copy_table(sObjs, scArg.objs, scArg.n_objs);
        shade_copy_table(sMtls, scArg.mtls, scArg.n_mtls);
        shade_copy_table(sTex, scArg.texmaps, scArg.n_texmaps);
        sc.objs = sObjs;
        sc.mtls = sMtls;
        sc.texmaps = sTex;
    }
    segq_load_table(qIn, qtab);
    __syncthreads();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool alive = false;
    uint32_t slot = 0;
    // n may be an upper bound (the host does not read every bounce's count back): the queue's own total decides
    if (i < n && i < qtab[QSEG]) {
        slot = queueIn[segq_item_index(qtab, i)];
        const float4 ra = pool.rays[2 * (size_t)slot], rb = pool.rays[2 * (size_t)slot + 1];
        const float4 hv = pool.hits[slot];
        const RayT ray{ra.x, ra.y, ra.z, rb.x, rb.y, rb.z, ra.w};
        const int hitSlot = __float_as_int(hv.w);
        // culled: a hit that the reference's box tests would have removed (lean extend kernel: not checked there) - the body
        // leaves the vertex alone and the path goes to the general tail kernel, which traces its ray again (shade_body.h)
        bool culled = false;
        // the lobe evaluation of the vertex's light samples is nee_kernel's (kernels_nee.hip): the body queues the pairs in qNee
        // and leaves the vertex's record at index i of pool.neeCtx
        const uint32_t ws = REPACK ? wbase + i : slot;
        const uint32_t os = REPACK ? path_origin(pool, slot) : 0u;
        alive = shade_body<MICRO, false, PHOT, PHONG, true>(sc, cam, pool, pm, slot, ray, __float_as_uint(rb.w), HitT{hv.x, hv.y, hv.z, hitSlot}, bounce, maxBounces,
                                                            seed, sBase, S, pixels, qShadow, i, &qNee, scArg.lights, ws, &culled);
        if (REPACK) {
            // (a culled vertex stays where it is: the general tail kernel finishes its path there)
            pool.origin[ws - pool.baseCap] = culled ? 0xFFFFFFFFu : os;
            if (!culled && slot >= pool.baseCap) pool.origin[slot - pool.baseCap] = 0xFFFFFFFFu;   // moved on from an earlier area
        }
        slot = ws;
    }
    // ---- surviving paths -> next queue ----
    __shared__ uint32_t qtmp[SEGQ_BLOCK_WORDS];
    const uint32_t q = segq_append_block(qOut, alive, 1u, gridDim.x, qtmp);
    if (alive) queueOut[q] = slot;
}

// End of a pass that moved paths (pathstate.h "Compaction"): the radiance of every path that finished in the compaction area goes
// back to the slot the path was generated in, with the NEE of its last vertex added as film_kernel would add it (resolve_add:
// the same additions in the same order), so that film_kernel finds every sample where ray generation put it.
__global__ void unpack_kernel(PathPool pool, uint32_t nUsed, int nLights, int usePhotons) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nUsed) return;
    const uint32_t o = pool.origin[j];
    if (o == 0xFFFFFFFFu) return;
    const uint32_t w = pool.baseCap + j;
    float4 L4 = pool.L[w];
    f3 L = mk3(L4.x, L4.y, L4.z);
    resolve_add(pool, w, nLights, usePhotons != 0, L);
    pool.L[o] = make_float4(L.x, L.y, L.z, L4.w);
    pool.pendBeta[o] = make_float4(0.0f, 0.0f, 0.0f, -1.0f);
}
void launch_unpack(const PathPool& pool, uint32_t nUsed, int nLights, int usePhotons, hipStream_t stream) {
    if (!nUsed) return;
    hipLaunchKernelGGL(unpack_kernel, dim3((nUsed + 255) / 256), dim3(256), 0, stream, pool, nUsed, nLights, usePhotons);
}

// ---- shadow --------------------------------------------------------------------------------------------------
// Visible() for every queued NEE segment (RayCaster.cpp:465): writes the transmittance next to its contribution.
// Same structure as the extend kernel (persistent waves, per-lane refill from the segmented queue, one inner or
// leaf step per iteration); a lane keeps its segment over the hops of the transmittance chain.
#ifndef LEAF_WEIGHT
#define LEAF_WEIGHT 1u
#endif
#ifndef SHADOW_ANYHIT
#define SHADOW_ANYHIT 1
#endif
// lanes that have ended a hop or hold no segment before the wave resolves / refills them: a shadow hop's end costs more than a
// closest-hit ray's (light spheres, the chain's next hop, the box-cull check) and is worth waiting for more lanes than the
// extend kernel's 16 - shadow per C3 frame at 8 / 16 / 24 / 32 / 40 / 48: 180 / 171 / 169 / 167 / 171 / 177 ms
#ifndef SHADOW_REFILL_MIN
#define SHADOW_REFILL_MIN 32
#endif
// useTransmittance: bit 0 = Visible() with transmittance (else VisibleSimple), bit 1 = no early exit (parity checks)
// CULL (lean kernel only): hops that end on a hit are checked against the reference's box tests (PathPool::cullShadows)
// Blocker cache (experiment, compiled in with -DRTX_BLOCKER_CACHE by tools/build_variant.sh; RTX_AMD_BLOCKER_CACHE=cells at run
// time): a hash table over cells of shadow-ray origins x lights that remembers the opaque triangle which last blocked a ray from
// that cell. Measured on the C3 frame (round 3): 45 % of the hops are blocked, but only 2.4 / 4.4 / 9.1 % of all hops find their
// blocker in the cache with cells of 1/128, 1/512, 1/2048 of the scene (the occluders of this scene are millimetre-sized cereal
// pieces: the triangle that blocks one ray seldom blocks its neighbour), and the candidate's fetch and test on the other hops
// cost more: shadow 184-186 ms per frame against 162 without the code (173 with the code compiled in and the table off).
// index of the entry for a shadow ray that starts at o towards light l:
__device__ __forceinline__ uint32_t blocker_key(const DevScene& sc, const PathPool& pool, f3 o, uint32_t l) {
    const int cx = (int)((o.x - sc.bounds[0]) * pool.blockerScale), cy = (int)((o.y - sc.bounds[1]) * pool.blockerScale),
              cz = (int)((o.z - sc.bounds[2]) * pool.blockerScale);
    uint32_t h = ((uint32_t)cx * 73856093u) ^ ((uint32_t)cy * 19349663u) ^ ((uint32_t)cz * 83492791u);
    h = (h ^ (h >> 15)) * 8u + l;
    return h & pool.blockerMask;
}
// COUNT: nodes visited and triangles tested are tallied (counters[3], [4]) - the instrumented build behind rtx_set_counters, which
// gives the bench the algorithmic bytes of a shadow hop (48 + nodes x 64 + triangles x 48)
// (the general kernel runs best at four waves per SIMD - 4 / 5 / 6: 64 / 85 / 134 ms on the C3 scene forced through it - and sits
// right at that register limit: it is told to stay there; the lean kernels are left to the compiler: 78-91 registers, five or six waves)
template <bool XF, bool CULL = false, bool COUNT = false, bool FRONT = false>
#ifdef SHADOW_WAVES_PER_EU
__attribute__((amdgpu_waves_per_eu(SHADOW_WAVES_PER_EU)))
#else
__attribute__((amdgpu_waves_per_eu(XF ? 4 : 6)))   // (lean kernel with the box-cull checks: eight registers spilled at six waves, and still 179 -> 171 ms per C3 frame against five)
#endif
__global__ __launch_bounds__(TRAV_BLOCK) void shadow_kernel(DevScene scArg, PathPool pool, SegQ q, int useTransmittance, uint2* spill,
                                                            uint32_t chunk, const uint32_t* __restrict__ items, unsigned long long* counters) {
    __shared__ uint2 lds_stack[STACK_LDS * TRAV_BLOCK];
    const DevSceneT<XF> sc(scArg);
    const uint32_t gtid = blockIdx.x * blockDim.x + threadIdx.x;
    TravStack st = make_trav_stack(&lds_stack[threadIdx.x], spill, gtid, gridDim.x * blockDim.x);
    uint32_t hops = 0, nodeCount = 0, triCount = 0;
#ifdef SHADOW_CACHE_STATS
    uint32_t cacheHits = 0;
#endif
    TravState s;
    s.cur = RTX_LANE_IDLE;   // a lane's state is its s.cur alone (extend_kernel): -1 a hop that has ended, -2 no segment in the lane
    s.sp = 0;
    bool exhausted = false;
    SegWalk walk;
    segq_walk_init<FRONT>(q, walk, gtid >> 6, chunk);
    uint32_t code = 0, entry = 0;
    f3 o = mk3(0.0f), d = mk3(0.0f);
    float maxDist = 0.0f, visibility = 1.0f;
    int hop = 0;
    // Early end of a hop, made exact: once an opaque blocker in front of the segment's end and of the lights is known, the
    // hop's answer is 0 unless a surface that lets light through is hit before it (the chain would then restart from an
    // offset origin - 1e-4 along that surface's normal, sideways to the ray - and may slip past the blocker's edge, or
    // past the blocker itself when the two are closer than the offset). So the traversal goes on, but only into subtrees
    // that hold such surfaces (mode 1: whole opaque subtrees are dropped unfetched); if one is hit in front of the
    // blocker, the hop is traced again as a plain closest hit (mode 2). Visible() is then the reference's bit for bit.
    int mode = 0;   // 0: looking for the first decisive blocker, 1: blocker known, 2: plain closest hit, 3: about to start 2
    for (;;) {
        // hops that reached their end are resolved in batches (see extend_kernel): the chain either ends or starts
        // its next hop in the same lane
        const bool fin = s.cur == RTX_TRAV_DONE;
        const unsigned long long finMask = __ballot(fin);
        const unsigned long long freeMask = finMask | __ballot(s.cur == RTX_LANE_IDLE);
        if (finMask != 0ull && ((uint32_t)__popcll(freeMask) >= SHADOW_REFILL_MIN || freeMask == ~0ull)) {
            // lean kernel: a hop that ends on a hit the reference's box tests would have culled (not checked on the way: see
            // ref_box_hit) is not resolved here - the whole chain is traced again by the general kernel (launch_shadow_culled)
            bool culledHop = false;
            // (measured: the objects' boxes in LDS and the best hit's object word kept in a register - no memory to wait for
            // here - cost more in the traversal loop than they save: 317 ms per C3 frame against 191)
            // (a triangle well inside both boxes - TRI_INTERIOR of the word kept from the triangle test, most of them - needs no fetch)
            if (!XF && CULL && fin && mode != 3 && pool.cullShadows && s.best.slot >= 0 && !ref_reaches_interior(s.bestWord, s.r, s.best.t) &&
                !ref_reaches_triangle<true>(scArg, s.r, s.best.slot, s.best.t)) {
                const uint32_t k = atomicAdd(&pool.totals[5], 1u);
                if (k < RTX_CULL_CAP) {
                    pool.cullShadows[k] = entry;
                    culledHop = true;
                }
            }
            if (culledHop) {
                hops -= (uint32_t)hop + 1u;   // counted again there
                s.cur = RTX_LANE_IDLE;
            } else if (fin && mode != 3 && !trav_verify(s, sc)) {
                mode = 0;   // general kernels: the hop's hit is culled by its object's box in the reference, traced again without that object
            } else if (fin) {
                bool next = true;
                if (mode != 3) {
                    intersect_lights(scArg, s.r, s.best);
                    float T;
                    if (shadow_hop_end(sc, o, d, maxDist, s.best, (useTransmittance & 1) != 0, hop, visibility, T, s.bestWord)) {
                        pool.nee[(size_t)(code & 7u) * pool.capacity + (code >> 3)].w = T;
                        s.cur = RTX_LANE_IDLE;
                        next = false;
                    } else
                        hops++;
                }
                if (next) {   // the next hop, or the same hop again as a plain closest hit (mode 3 -> 2)
                    trav_init(s, sc, RayT{o.x, o.y, o.z, d.x, d.y, d.z, maxDist});
                    mode = mode == 3 ? 2 : 0;
                }
            }
        }
        const unsigned long long idle = __ballot(s.cur == RTX_LANE_IDLE);
        const uint32_t nIdle = (uint32_t)__popcll(idle);
        if (!exhausted && nIdle >= SHADOW_REFILL_MIN) {
            uint32_t first = 0;
            const uint32_t got = segq_take<FRONT>(q, walk, nIdle, first);
            if (got == 0u) exhausted = true;
            if (s.cur == RTX_LANE_IDLE) {
                const uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                if (prefix < got) {
                    entry = (XF && items) ? items[first + prefix] : first + prefix;   // (a list: launch_shadow_culled)
                    const size_t e = entry;
                    const float4 a = pool.shadowQ[2 * e], b = pool.shadowQ[2 * e + 1];
                    code = __float_as_uint(b.w);
                    o = mk3(a.x, a.y, a.z);
                    d = mk3(b.x, b.y, b.z);
                    maxDist = a.w;
                    visibility = 1.0f;
                    hop = 0;
                    trav_init(s, sc, RayT{o.x, o.y, o.z, d.x, d.y, d.z, maxDist});
                    mode = 0;
                    hops++;
                    // Blocker cache (lean scenes): the opaque triangle that last blocked a ray from this neighbourhood towards this
                    // light is tested before anything else. If it blocks this ray too - a hit that decides the hop,
                    // shadow_hit_decides - the lane starts where a traversal would have arrived after finding it: blocker known,
                    // only subtrees with surfaces that let light through are left to look at (mode 1). The hop's answer is formed
                    // by the same rules from there on, so the cache changes no result, whatever it holds.
#ifdef RTX_BLOCKER_CACHE   // (variant builds only: measured not to pay on this scene, see the comment at blocker_key)
                    if (!XF && pool.blockerCache && (useTransmittance & 2) == 0) {
                        const int cand = pool.blockerCache[blocker_key(scArg, pool, o, code & 7u)];
                        if (cand >= 0) {
                            const float4* tris4 = reinterpret_cast<const float4*>(sc.tris);
                            const float4 ta = tris4[(size_t)cand * 3 + 0], tb = tris4[(size_t)cand * 3 + 1], tc = tris4[(size_t)cand * 3 + 2];
                            test_triangle(ta, tb, tc, s.r, RTX_HIT_FRONT_AND_BACK, (uint32_t)cand, sc.tris, s.best, s.bestPrim, s.bestWord);
                            if (s.best.slot >= 0 && shadow_hit_decides(scArg, s, maxDist, (useTransmittance & 1) != 0)) {
#ifdef SHADOW_CACHE_STATS
                                cacheHits++;
#endif
                                if ((useTransmittance & 1) != 0) {
                                    mode = 1;
                                    trav_skip_opaque(s, st);
                                } else
                                    s.cur = RTX_TRAV_DONE;   // VisibleSimple: any surface in front of the segment's end blocks
                            } else
                                trav_init(s, sc, RayT{o.x, o.y, o.z, d.x, d.y, d.z, maxDist});   // no help: a plain hop
                        }
                    }
#endif
                }
            }
        }
        {
            const bool atInner = (int)s.cur >= 0;
            const bool atLeaf = (int)s.cur < (int)RTX_LANE_IDLE;
            const unsigned long long innerMask = __ballot(atInner), leafMask = __ballot(atLeaf);
            if ((innerMask | leafMask) == 0ull) {
                // nothing to step: hops that have ended are resolved at the top of the loop, a drained queue ends it
                if (exhausted && __ballot(s.cur == RTX_TRAV_DONE) == 0ull) break;
                continue;
            }
            const uint32_t nInner = (uint32_t)__popcll(innerMask), nLeaf = (uint32_t)__popcll(leafMask);
            if (nLeaf * LEAF_WEIGHT >= nInner) {
                if (atLeaf) {
#if SHADOW_ANYHIT
                    const int before = s.best.slot;
                    const uint32_t nt = trav_leaf_test(s, sc, s.cur, RTX_HIT_FRONT_AND_BACK);
                    if (COUNT) triCount += nt;
                    bool stop = false;   // the hop ends here: decided (VisibleSimple), or to be traced again
                    if (s.best.slot != before && (useTransmittance & 2) == 0 && mode != 2) {
                        const bool withT = (useTransmittance & 1) != 0;
                        if (mode == 1) {
                            // a closer hit: another opaque surface changes nothing, one that lets light through ends the shortcut
                            if ((s.bestWord & TRI_OPAQUE) == 0u) {
                                mode = 3;
                                stop = true;
                            }
                        } else if (shadow_hit_decides(scArg, s, maxDist, withT)) {
                            if (withT) mode = 1;
                            else stop = true;   // VisibleSimple: whatever the closest hit is, it lies in front of this one and blocks
#ifdef RTX_BLOCKER_CACHE
                            if (!XF && pool.blockerCache && hop == 0) pool.blockerCache[blocker_key(scArg, pool, o, code & 7u)] = s.best.slot;
#endif
                        }
                    }
                    if (stop) {
                        s.cur = RTX_TRAV_DONE;
                        s.sp = 0;
                    } else if (mode == 1) {   // blocker known: only subtrees with surfaces that let light through
                        s.cur = 0u;           // (an untagged reference: the leaf just tested is done with)
                        trav_skip_opaque(s, st);
                    } else
                        trav_pop(s, st);
#else
                    trav_leaf_step(s, sc, st, RTX_HIT_FRONT_AND_BACK);
#endif
                }
            } else if (atInner) {
                if (COUNT) nodeCount++;
                trav_inner_step(s, sc, st);
#if SHADOW_ANYHIT
                if (mode == 1) trav_skip_opaque(s, st);
#endif
            }
        }
    }
    atomicAdd(&pool.rayCounters[0], (unsigned long long)hops);
    if (COUNT) {
        atomicAdd(&counters[3], (unsigned long long)nodeCount);
        atomicAdd(&counters[4], (unsigned long long)triCount);
    }
#ifdef SHADOW_CACHE_STATS
    atomicAdd(&pool.rayCounters[4], (unsigned long long)cacheHits);
#endif
}

// ---- film (RayCaster.cpp:798-838) -----------------------------------------------------------------------------
__device__ __forceinline__ float filter_radius(int type, float radiusOverride) {   // :84-92, :807
    if (radiusOverride > 0.0f) return radiusOverride;
    switch (type) {
        case RTX_FILTER_GAUSSIAN: return 1.5f;
        case RTX_FILTER_MITCHELL: return 2.0f;
        case RTX_FILTER_LANCZOS: return 2.0f;
        default: return 1.0f;
    }
}
__device__ __forceinline__ float filter1d(int type, float radius, float x) {   // :643-690
    switch (type) {
        case RTX_FILTER_GAUSSIAN: {
            float sigma = radius * 0.5f;
            if (sigma <= 0.0f) sigma = 0.5f;
            const float t = x / sigma;
            return rtx_expf(-0.5f * t * t);
        }
        case RTX_FILTER_MITCHELL: {
            const float B = 1.0f / 3.0f;
            const float C = 1.0f / 3.0f;
            x = fabsf(x);
            if (x < 1.0f)
                return ((12.0f - 9.0f * B - 6.0f * C) * x * x * x + (-18.0f + 12.0f * B + 6.0f * C) * x * x + (6.0f - 2.0f * B)) / 6.0f;
            else if (x < 2.0f)
                return ((-B - 6.0f * C) * x * x * x + (6.0f * B + 30.0f * C) * x * x + (-12.0f * B - 48.0f * C) * x + (8.0f * B + 24.0f * C)) / 6.0f;
            return 0.0f;
        }
        case RTX_FILTER_LANCZOS: {
            const float a = radius;
            x = fabsf(x);
            if (x >= a) return 0.0f;
            if (x < 1e-6f) return 1.0f;
            const float pix = PI_F * x;
            return (a * rtx_sinf(pix) * rtx_sinf(pix / a)) / (pix * pix);
        }
        default: {
            x = fabsf(x);
            if (x >= radius) return 0.0f;
            return 1.0f - x / radius;
        }
    }
}

// One lane per pixel of the pass: walks its S samples in sample order (so per-pixel sums and the running z are
// accumulated in the reference's order) and splats them into the film with float atomics.
//
// NEAR3 (tent filter with radius <= 1, the reference's default): every sample of a pixel lies inside that pixel, so
// it can only reach the 3x3 pixels around it. The lane sums the S samples' weighted contributions for those nine
// pixels in registers and issues the atomics once at the end: 9 x 5 atomics per pixel and pass instead of per sample.
template <bool NEAR3>
__global__ void film_kernel(PathPool pool, rtx_film film, const int* __restrict__ pixels, uint32_t nPix, int S, int W, int H,
                            int filterType, float filterRadius, int resetZ, float4* __restrict__ samplesOut, int sBase,
                            int sppTotal, float radianceClamp, int nLights, int usePhotons) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nPix) return;
    const int pix = pixels[i];
    if (pix < 0) return;
    const uint32_t chunk = i >> 6, lane = i & 63u;
    float4 s1 = pool.statSum[pix];
    float4 s2 = pool.statSum2[pix];
    int n = __float_as_int(s1.w);
    float bestZ = (resetZ || n == 0) ? RTX_BIG : s2.w;   // zero-filled statistics are a valid start
    const float radius = filter_radius(filterType, filterRadius);
    int cnt = film.count[pix];
    bool guide = film.albedo ? (film.albedo[4 * (size_t)pix + 3] != 0.0f) : true;
    const int px = pix % W, py = pix / W;
    float accR[9], accG[9], accB[9], accW[9], accZ[9];
    if (NEAR3) {
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            accR[k] = accG[k] = accB[k] = accW[k] = 0.0f;
            accZ[k] = RTX_BIG;
        }
    }
    for (int s = 0; s < S; ++s) {
        const uint32_t slot = (chunk * (uint32_t)S + (uint32_t)s) * 64u + lane;
        float4 L4 = pool.L[slot];
        {   // a path that ended in the bounce loop still owes the NEE of its last vertex (resolve_add)
            f3 Lr = mk3(L4.x, L4.y, L4.z);
            resolve_add(pool, slot, nLights, usePhotons != 0, Lr);
            L4.x = Lr.x; L4.y = Lr.y; L4.z = Lr.z;
        }
        // kRadianceClamp (RayCaster.cpp:569-571) in path mode; TraceLegacy returns its radiance unclamped
        const float cr = fminf(L4.x, radianceClamp), cg = fminf(L4.y, radianceClamp), cb = fminf(L4.z, radianceClamp);
        if (samplesOut) samplesOut[(size_t)pix * sppTotal + sBase + s] = make_float4(cr, cg, cb, L4.w);
        if (L4.w < bestZ) bestZ = L4.w;
        cnt += 1;
        if (!guide) {
            const float4 al = pool.albedo[slot];
            if (al.w != 0.0f) {
                guide = true;
                float4 nn = pool.normal[slot];
                {   // StoreGuide normalises the path's (already unit) normal once more (RayCaster.cpp:713)
                    const f3 n2 = safe_normalize3(mk3(nn.x, nn.y, nn.z));
                    nn.x = n2.x; nn.y = n2.y; nn.z = n2.z;
                }
                float* A = film.albedo + 4 * (size_t)pix;
                A[0] = al.x; A[1] = al.y; A[2] = al.z; A[3] = 1.0f;
                if (film.normal) {
                    float* Nn = film.normal + 4 * (size_t)pix;
                    Nn[0] = nn.x; Nn[1] = nn.y; Nn[2] = nn.z; Nn[3] = 1.0f;
                }
            }
        }
        const float2 sp = pool.samplePos[slot];
        if (NEAR3) {
            float wx[3], wy[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                wx[k] = filter1d(RTX_FILTER_TENT, radius, (float)(px + k - 1) + 0.5f - sp.x);
                wy[k] = filter1d(RTX_FILTER_TENT, radius, (float)(py + k - 1) + 0.5f - sp.y);
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float w = wx[kx] * wy[ky];
                    if (wx[kx] > 0.0f && wy[ky] > 0.0f) {
                        const int k = ky * 3 + kx;
                        accR[k] += cr * w;
                        accG[k] += cg * w;
                        accB[k] += cb * w;
                        accW[k] += w;
                        accZ[k] = fminf(accZ[k], bestZ);
                    }
                }
        } else {
            const int x0 = max(0, (int)floorf(sp.x - radius));
            const int x1 = min(W - 1, (int)ceilf(sp.x + radius));
            const int y0 = max(0, (int)floorf(sp.y - radius));
            const int y1 = min(H - 1, (int)ceilf(sp.y + radius));
            for (int yy = y0; yy <= y1; ++yy) {
                const float wy = filter1d(filterType, radius, (float)yy + 0.5f - sp.y);
                if (wy <= 0.0f) continue;
                for (int xx = x0; xx <= x1; ++xx) {
                    const float wx = filter1d(filterType, radius, (float)xx + 0.5f - sp.x);
                    if (wx <= 0.0f) continue;
                    const float w = wx * wy;
                    const size_t idx = (size_t)yy * W + xx;
                    atomicAdd(&film.accum[4 * idx + 0], cr * w);
                    atomicAdd(&film.accum[4 * idx + 1], cg * w);
                    atomicAdd(&film.accum[4 * idx + 2], cb * w);
                    atomicAdd(&film.accum[4 * idx + 3], w);
                    atomicMin(reinterpret_cast<int*>(&film.z[idx]), __float_as_int(bestZ));   // z >= 0: int order == float order
                }
            }
        }
        n += 1;
        s1.x += cr; s1.y += cg; s1.z += cb;
        s2.x += cr * cr; s2.y += cg * cg; s2.z += cb * cb;
    }
    if (NEAR3) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int k = ky * 3 + kx;
                const int xx = px + kx - 1, yy = py + ky - 1;
                if (accW[k] > 0.0f && xx >= 0 && xx < W && yy >= 0 && yy < H) {
                    const size_t idx = (size_t)yy * W + xx;
                    atomicAdd(&film.accum[4 * idx + 0], accR[k]);
                    atomicAdd(&film.accum[4 * idx + 1], accG[k]);
                    atomicAdd(&film.accum[4 * idx + 2], accB[k]);
                    atomicAdd(&film.accum[4 * idx + 3], accW[k]);
                    atomicMin(reinterpret_cast<int*>(&film.z[idx]), __float_as_int(accZ[k]));
                }
            }
    }
    film.count[pix] = cnt;
    s1.w = __int_as_float(n);
    s2.w = bestZ;
    pool.statSum[pix] = s1;
    pool.statSum2[pix] = s2;
}

__global__ void fill_int_kernel(int* p, uint32_t n, int v) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ void film_clear_kernel(rtx_film film, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    reinterpret_cast<float4*>(film.accum)[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    film.count[i] = 0;
    film.z[i] = RTX_BIG;
    if (film.albedo) reinterpret_cast<float4*>(film.albedo)[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (film.normal) reinterpret_cast<float4*>(film.normal)[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}
__global__ void stats_clear_kernel(PathPool pool, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    pool.statSum[i] = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(0));
    pool.statSum2[i] = make_float4(0.0f, 0.0f, 0.0f, RTX_BIG);
}

// ---- launch wrappers -------------------------------------------------------------------------------------------
uint32_t launch_raygen(const DevCamera& cam, const PathPool& pool, const int* pixels, uint32_t nPix, int S, int sBase, int indexStride,
                       const SegQ& qOut, uint32_t* queueOut, int initGuides, hipStream_t stream) {
    const uint32_t n = nPix * (uint32_t)S;
    if (!n) return 0;
    const uint32_t grid = (n + 255) / 256;
    hipLaunchKernelGGL(raygen_kernel, dim3(grid), dim3(256), 0, stream, cam, pool, pixels, nPix, S, sBase, indexStride, qOut, queueOut,
                       initGuides);
    return grid;
}
// repackBase: 0 = paths stay in their slots; else the first slot of the compaction area this bounce writes to (>= pool.baseCap)
uint32_t launch_shade(const DevScene& sc, const DevCamera& cam, const PathPool& pool, const DevPhotonMap& pm, const SegQ& qIn,
                      const uint32_t* queueIn, uint32_t n, int bounce, int maxBounces, unsigned long long seed, int sBase, int S,
                      const int* pixels, const SegQ& qOut, uint32_t* queueOut, const SegQ& qShadow, int materialKinds,
                      hipStream_t stream, uint32_t repackBase) {
    if (!n) return 0;
    const uint32_t grid = (n + RTX_SHADE_BLOCK - 1) / RTX_SHADE_BLOCK;
    const bool ldsTab = sc.n_objs <= SHADE_LDS_OBJS && sc.n_mtls <= SHADE_LDS_MTLS && sc.n_texmaps <= SHADE_LDS_TEX;
#define RTX_LAUNCH_SHADE_R(M, P, G, T, R)                                                                                            \
    hipLaunchKernelGGL((shade_kernel<M, P, G, T, R>), dim3(grid), dim3(RTX_SHADE_BLOCK), 0, stream, sc, cam, pool, pm, qIn, queueIn, n, bounce, maxBounces, \
                       seed, sBase, S, pixels, qOut, queueOut, qShadow, pool.qN, repackBase)
#define RTX_LAUNCH_SHADE(M, P, G)                                                                                                    \
    if (ldsTab) {                                                                                                                    \
        if (repackBase) RTX_LAUNCH_SHADE_R(M, P, G, true, true);                                                                     \
        else RTX_LAUNCH_SHADE_R(M, P, G, true, false);                                                                               \
    } else {                                                                                                                         \
        if (repackBase) RTX_LAUNCH_SHADE_R(M, P, G, false, true);                                                                    \
        else RTX_LAUNCH_SHADE_R(M, P, G, false, false);                                                                              \
    }
    // materialKinds: bit 0 = the scene holds MtlMicrofacet, bit 1 = MtlPhong; a caustics map selects the gather
    switch ((materialKinds & 3) | (pm.n > 0 ? 4 : 0)) {
        case 0: RTX_LAUNCH_SHADE(false, false, false); break;
        case 1: RTX_LAUNCH_SHADE(true, false, false); break;
        case 2: RTX_LAUNCH_SHADE(false, false, true); break;
        case 3: RTX_LAUNCH_SHADE(true, false, true); break;
        case 4: RTX_LAUNCH_SHADE(false, true, false); break;
        case 5: RTX_LAUNCH_SHADE(true, true, false); break;
        case 6: RTX_LAUNCH_SHADE(false, true, true); break;
        default: RTX_LAUNCH_SHADE(true, true, true); break;
    }
#undef RTX_LAUNCH_SHADE
#undef RTX_LAUNCH_SHADE_R
    return grid;
}
void launch_shadow(const DevScene& sc, const LaunchCfg& cfg, const PathPool& pool, const SegQ& q, uint32_t n, int useTransmittance) {
    if (!n) return;
    const uint32_t need = (n + TRAV_BLOCK - 1) / TRAV_BLOCK;
    const uint32_t grid = need < (uint32_t)cfg.gridBlocks ? need : (uint32_t)cfg.gridBlocks;
#define RTX_LAUNCH_SHADOW(X, C, N)                                                                                                          \
    hipLaunchKernelGGL((shadow_kernel<X, C, N>), dim3(grid), dim3(TRAV_BLOCK), 0, cfg.stream, sc, pool, q, useTransmittance, cfg.spill,     \
                       segq_chunk(n, grid * (TRAV_BLOCK / 64)), (const uint32_t*)nullptr, cfg.counters)
#define RTX_LAUNCH_SHADOW_FRONT(X, C)                                                                                                       \
    hipLaunchKernelGGL((shadow_kernel<X, C, false, true>), dim3(grid), dim3(TRAV_BLOCK), 0, cfg.stream, sc, pool, q, useTransmittance,      \
                       cfg.spill, segq_chunk(n, grid * (TRAV_BLOCK / 64)), (const uint32_t*)nullptr, cfg.counters)
    if (!cfg.counting && segq_front(n)) {   // a launch of middling size walks its queue as one front (segqueue.h SegWalk)
        if (sc.has_xf) RTX_LAUNCH_SHADOW_FRONT(true, false);
        else if (pool.cullShadows) RTX_LAUNCH_SHADOW_FRONT(false, true);
        else RTX_LAUNCH_SHADOW_FRONT(false, false);
        return;
    }
    if (sc.has_xf) {
        if (cfg.counting) RTX_LAUNCH_SHADOW(true, false, true);
        else RTX_LAUNCH_SHADOW(true, false, false);
    } else if (pool.cullShadows)
        RTX_LAUNCH_SHADOW(false, true, false);   // (the box-cull checks and the counters exclude each other: rtx_render.cpp)
    else if (cfg.counting)
        RTX_LAUNCH_SHADOW(false, false, true);
    else
        RTX_LAUNCH_SHADOW(false, false, false);
#undef RTX_LAUNCH_SHADOW
#undef RTX_LAUNCH_SHADOW_FRONT
}
// The chains the lean shadow kernel set aside (pool.cullShadows, their number in pool.totals[5]) through the general
// kernel. The host does not know the number: a one-wave kernel turns it into an identity queue over the list and resets it,
// and a two-block grid drains that queue (a frame holds a few dozen such chains).
__global__ void segq_identity_counted_kernel(SegQ q, uint32_t* count, uint32_t cap) {
    const uint32_t l = threadIdx.x;
    const uint32_t n = min(*count, cap);
    const uint32_t lo = (uint32_t)(((unsigned long long)l * n) / QSEG);
    q.ctl[SEGQ_PREFIX + l] = lo;
    q.ctl[SEGQ_START + l] = lo;
    q.ctl[SEGQ_COUNT + l * QSTRIDE] = 0u;
    q.ctl[SEGQ_CURSOR + l * QSTRIDE] = 0u;
    __syncthreads();
    if (l == 63u) {
        q.ctl[SEGQ_PREFIX + QSEG] = n;
        *count = 0u;
    }
}
void launch_shadow_culled(const DevScene& scGeneral, const LaunchCfg& cfg, const PathPool& pool, const SegQ& qList, int useTransmittance) {
    hipLaunchKernelGGL(segq_identity_counted_kernel, dim3(1), dim3(64), 0, cfg.stream, qList, pool.totals + 5, RTX_CULL_CAP);
    hipLaunchKernelGGL((shadow_kernel<true, false, false>), dim3(2), dim3(TRAV_BLOCK), 0, cfg.stream, scGeneral, pool, qList, useTransmittance, cfg.spill, 64u,
                       (const uint32_t*)pool.cullShadows, cfg.counters);
}
void launch_film(const PathPool& pool, const rtx_film& film, const int* pixels, uint32_t nPix, int S, int W, int H, int filterType,
                 float filterRadius, int resetZ, float4* samplesOut, int sBase, int sppTotal, float radianceClamp, int nLights,
                 int usePhotons, hipStream_t stream) {
    if (!nPix) return;
    const bool near3 = filterType == RTX_FILTER_TENT && (filterRadius <= 0.0f || filterRadius <= 1.0f);
    if (near3)
        hipLaunchKernelGGL(film_kernel<true>, dim3((nPix + 255) / 256), dim3(256), 0, stream, pool, film, pixels, nPix, S, W, H,
                           filterType, filterRadius, resetZ, samplesOut, sBase, sppTotal, radianceClamp, nLights, usePhotons);
    else
        hipLaunchKernelGGL(film_kernel<false>, dim3((nPix + 255) / 256), dim3(256), 0, stream, pool, film, pixels, nPix, S, W, H,
                           filterType, filterRadius, resetZ, samplesOut, sBase, sppTotal, radianceClamp, nLights, usePhotons);
}
void launch_fill_int(int* p, uint32_t n, int v, hipStream_t stream) {
    if (!n) return;
    hipLaunchKernelGGL(fill_int_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, p, n, v);
}
void launch_film_clear(const rtx_film& film, uint32_t n, hipStream_t stream) {
    hipLaunchKernelGGL(film_clear_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, film, n);
}
void launch_stats_clear(const PathPool& pool, uint32_t n, hipStream_t stream) {
    hipLaunchKernelGGL(stats_clear_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, pool, n);
}
