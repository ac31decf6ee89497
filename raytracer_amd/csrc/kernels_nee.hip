// nee_kernel: the lobe evaluation of the light loop of TracePath (reference RayTracer/RayCaster.cpp:456-503) for the
// vertices a shade launch has just shaded, one lane per (vertex, light) pair instead of one lane per vertex.
//
// Why a kernel of its own. Inside shade_kernel the loop ran for every lane of a wave three times over (cerealBar.xml has
// three lights), yet only one (vertex, light) pair in six ends in a shadow ray: half of the samples on a sphere light face
// away from the vertex (PointLight::GenerateSample rejects them, Shading.cpp:150-152), half of the rest lie behind an
// opaque surface, and what is left is the expensive part - GetSampleInfo with its powf (Shading.cpp:533-713) and the MIS
// weight. A wave spent most of those instructions on a minority of its lanes (28 of 64 active per VALU instruction over the
// whole kernel, profiles/r02/pmc_summary_spp8_final.txt). Now the work is cut in two where the lanes thin out:
//   A  (shade_body.h, in shade_kernel, every lane with a surface) the two random numbers, the point on the light, the cheap
//      rejections. Pairs that are left - about one in four - are appended to the NEE queue with the sampled direction, and
//      the vertex leaves its lobe terms in a record (nee_ctx.h) if it has such a pair.
//   B  (here) the queued pairs, packed densely over the lanes whatever vertex and light they belong to: GetSampleInfo, MIS
//      weight, contribution, and the shadow-queue entry for the shadow kernel.
// The arithmetic of a pair is the light loop's, operation for operation (bsdf.h), and its random numbers are counter-based
// (vec.h), so the per-sample radiance is the same bits as with the loop inside the shading body - which the tail kernel
// still runs (shade_body.h, SPLIT = false); the parity tests cover both.
#include "kernels.h"
#include "pathstate.h"
#include "bsdf.h"
#include "shadow.h"
#include "segqueue.h"
#include "shade_body.h"

#define NEE_BLOCK 256u
#ifndef NEE_WAVES_PER_EU
#define NEE_WAVES_PER_EU 4
#endif
// nBound: the queue's entries, or an upper bound of them (vertices x lights when the caller does not know the count: blocks
// beyond the true total leave at once - at a price, 7 ms per C3 frame for three empty blocks in four: rtx_render.cpp reads the
// count back).
template <bool MICRO, bool PHONG>
__global__ __launch_bounds__(NEE_BLOCK) __attribute__((amdgpu_waves_per_eu(NEE_WAVES_PER_EU))) void nee_kernel(DevScene sc, PathPool pool, SegQ qIn, uint32_t nBound, SegQ qShadow) {
    __shared__ uint32_t qtab[SEGQ_TABLE_WORDS];
    // The pair queue's segments are interleaved (SegQ::run: shade_kernel's blocks in flight append through 64 counters), so its
    // flat order goes 64 times over the vertices of the bounce. The blocks therefore take the flat chunks in transposed order -
    // block j the chunk (j % 64) * rows + j / 64, about the same place of segment j % 64 as its neighbours take of theirs -
    // and the shadow queue, which keeps the order of these blocks, holds the segments in the order of the vertices again.
    const uint32_t total = qIn.ctl[SEGQ_PREFIX + QSEG];
    const uint32_t rows = segq_nee_rows(total, NEE_BLOCK);
    const uint32_t chunkIdx = (blockIdx.x % QSEG) * rows + blockIdx.x / QSEG;   // (in flat order: shadow 153 -> 159 ms per C3 frame)
    if (blockIdx.x >= QSEG * rows || chunkIdx * NEE_BLOCK >= total) return;   // (block-uniform)
    segq_load_table(qIn, qtab);
    __syncthreads();
    const uint32_t i = chunkIdx * NEE_BLOCK + threadIdx.x;
    const bool valid = i < nBound && i < qtab[QSEG];
    const size_t stride = pool.neeStride;
    bool need = false;
    f3 c = mk3(0.0f), so = mk3(0.0f), sd = mk3(0.0f);
    float shadowMax = 0.0f;
    uint32_t vslot = 0u, l = 0u;
    if (valid) {
        const size_t e = segq_item_index(qtab, i);
        const float4 dd = pool.neeQ[2 * e], ee = pool.neeQ[2 * e + 1];
        const f3 ldir = mk3(dd.x, dd.y, dd.z);
        const uint32_t vflags = __float_as_uint(ee.z);   // NEEF_ flags | light << 8 | material << 16 (nee_ctx.h)
        const float alpha = ee.w;
        l = (vflags >> 8) & 0xFFu;
        vslot = __float_as_uint(ee.y);
        const float4* rec = pool.neeCtx + __float_as_uint(ee.x);
        const float4 w0 = NEE_W(rec, 0), w3 = NEE_W(rec, 3);
        const f3 vP = mk3(w0.x, w0.y, w0.z), GN = mk3(w3.x, w3.y, w3.z);
        const DevLight& light = sc.lights[l];
        DirInfo li;
        light_sample_terms(light, dd.w, li);
        shadowMax = (li.dist > 0.0f && li.dist < RTX_BIG * 0.5f) ? li.dist : RTX_BIG;
        SampCtx sx;
        sx.front = (vflags & NEEF_FRONT) != 0u;
        DirInfo bs;
        if (MICRO && (vflags & NEEF_MICRO) != 0u) {
            const MicroCtx mctx = nee_ctx_load_micro(rec, stride, alpha);
            evaluate_microfacet(mctx, sx, safe_normalize3(ldir), bs);
        } else {
            const bool phong = PHONG && (vflags & NEEF_PHONG) != 0u;
            PhongCtx ctx;
            if (vflags & NEEF_COMPACT) ctx = nee_ctx_rebuild(sc.mtls[vflags >> 16], w0, NEE_W(rec, 2), w3, sx.front, phong);
            else ctx = nee_ctx_load(rec, stride, alpha, vflags, phong);
            evaluate_phong_blinn(ctx, phong, sx, safe_normalize3(ldir), bs);
        }
        if (bs.prob > 0.0f && max3c(bs.mult) > 0.0f) {   // RayCaster.cpp:476-501
            if (light.size > 0.0f) {
                const float dist = li.dist;
                const float cosLight = light_cos_for_nee(light, vP, safe_normalize3(ldir));
                const float pdfSolidAngle = (cosLight > 1e-6f && dist > 1e-6f) ? li.prob * dist * dist / cosLight : 0.0f;
                if (pdfSolidAngle > 0.0f) {
                    const float w = power_heuristic(pdfSolidAngle, bs.prob);
                    const float G = 1.0f / fmaxf(1e-6f, dist * dist);
                    c = li.mult * bs.mult * G * (w / fmaxf(1e-6f, li.prob));
                }
            } else {
                const float w = power_heuristic(li.prob, bs.prob);
                c = li.mult * bs.mult * (w / fmaxf(1e-4f, li.prob));
            }
        }
        // the shadow segment is needed when it can contribute, or when its visibility scales the caustic term (:468, :518)
        need = max3c(c) > 0.0f || (vflags & NEEF_CAUSTIC) != 0u;
        so = offset_point(vP, GN, ldir);
        sd = safe_normalize3(ldir);
    }
    // regions of the shadow queue by the blocks that hold pairs, not by the grid (an upper bound): segq_append_g
    __shared__ uint32_t qtmp[SEGQ_BLOCK_WORDS];
    const uint32_t qs = segq_append_block(qShadow, need, 1u, QSEG * rows, qtmp);
    if (need) {
        pool.shadowQ[2 * (size_t)qs] = make_float4(so.x, so.y, so.z, shadowMax);
        pool.shadowQ[2 * (size_t)qs + 1] = make_float4(sd.x, sd.y, sd.z, __uint_as_float(vslot * 8u + l));
    }
    if (valid) pool.nee[(size_t)l * pool.capacity + vslot] = need ? make_float4(c.x, c.y, c.z, 0.0f) : make_float4(0.0f, 0.0f, 0.0f, -1.0f);
}

// The pairs the shade launch over n vertices has just queued in pool.qN (scanned). Returns the grid size: the geometry the
// scan of the shadow queue needs (one entry per thread at most).
uint32_t launch_nee(const DevScene& sc, const PathPool& pool, uint32_t n, const SegQ& qShadow, int materialKinds, hipStream_t stream, uint32_t pairs) {
    if (!n || sc.n_lights <= 0) return 0;
    const unsigned long long bound = pairs ? (unsigned long long)pairs : (unsigned long long)n * (unsigned long long)sc.n_lights;
    const uint32_t nBound = (uint32_t)(bound > 0xFFFFFF00ull ? 0xFFFFFF00ull : bound);   // (the pool holds fewer than 2^32 shadow segments: rtx_render.cpp)
    const uint32_t grid = QSEG * segq_nee_rows(nBound, NEE_BLOCK);   // (covers the kernel's own QSEG x rows, which comes from the true count)
#define RTX_LAUNCH_NEE(M, G) hipLaunchKernelGGL((nee_kernel<M, G>), dim3(grid), dim3(NEE_BLOCK), 0, stream, sc, pool, pool.qN, nBound, qShadow)
    switch (materialKinds & 3) {
        case 0: RTX_LAUNCH_NEE(false, false); break;
        case 1: RTX_LAUNCH_NEE(true, false); break;
        case 2: RTX_LAUNCH_NEE(false, true); break;
        default: RTX_LAUNCH_NEE(true, true); break;
    }
#undef RTX_LAUNCH_NEE
    return grid;
}
