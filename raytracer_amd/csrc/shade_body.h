// The per-vertex body of the path loop (one bounce of TracePath, reference RayTracer/RayCaster.cpp:359-567), shared by
// shade_kernel (kernels_path.hip: one launch per bounce, NEE visibility through the shadow queue) and tail_kernel
// (kernels_tail.hip: the last few paths of a pass walked to their ends inside one kernel).
#pragma once
#include "kernels.h"
#include "pathstate.h"
#include "bsdf.h"
#include "shadow.h"
#include "photon_gather.h"
#include "segqueue.h"
#include "nee_ctx.h"

// ---- shade ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ f3 apply_absorption(f3 c, f3 absorption, float dist) {   // RayCaster.cpp:281-301
    if (dist <= 0.0f) return c;
    if (absorption.x == 0.0f && absorption.y == 0.0f && absorption.z == 0.0f) return c;
    const float ar = rtx_expf(-fmaxf(0.0f, absorption.x) * dist);
    const float ag = rtx_expf(-fmaxf(0.0f, absorption.y) * dist);
    const float ab = rtx_expf(-fmaxf(0.0f, absorption.z) * dist);
    return mk3(fminf(c.x, c.x * ar), fminf(c.y, c.y * ag), fminf(c.z, c.z * ab));
}
__device__ __forceinline__ void dir_to_screen_uv(const DevCamera& cam, f3 dir, float& u, float& v) {   // :303-330
    const f3 f = ld3(cam.f), r = ld3(cam.r), uvec = ld3(cam.u);
    float dz = dot3(dir, f);
    if (fabsf(dz) < 1e-6f) dz = (dz >= 0.0f) ? 1e-6f : -1e-6f;
    const float sx = dot3(dir, r) / dz;
    const float sy = dot3(dir, uvec) / dz;
    const float fovRad = cam.fov * PI_F / 180.0f;
    const float halfY = rtx_tanf(0.5f * fovRad);
    const float aspect = (cam.H > 0) ? (float)cam.W / (float)cam.H : 1.0f;
    const float halfX = halfY * aspect;
    const float ndcX = sx / halfX;
    const float ndcY = sy / halfY;
    u = (ndcX + 1.0f) * 0.5f;
    v = (1.0f - ndcY) * 0.5f;
    u = u - floorf(u);
    v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
}
__device__ __forceinline__ float power_heuristic(float pdfA, float pdfB) {   // :102-106
    const float a = pdfA * pdfA;
    const float b = pdfB * pdfB;
    return a / (a + b);
}
__device__ __forceinline__ bool has_map_or_value(const float* c, int tex) {   // :333-337
    return tex >= 0 || c[0] != 0.0f || c[1] != 0.0f || c[2] != 0.0f;
}


// One bounce of TracePath (RayCaster.cpp:359-567) for every path in queueIn. Paths that continue are appended to
// queueOut; NEE shadow segments are appended to the shadow queue; their results are folded in by resolve_add.
// L += beta * sum_i c_i T_i and the caustic term of one path vertex once its shadow results are in pool.nee[].w
// (RayCaster.cpp:470-522), in the reference's order of additions into L. There is no resolve launch: the additions of
// bounce k are made by whoever reads the path's L next - the shading of bounce k+1 (they come before that vertex's own
// terms, as in the reference), the tail kernel, or the film kernel for a path that ended at bounce k.
// Returns false when nothing is pending (pendBeta.w < 0); otherwise pendBeta.w is the set of lights (bit l) whose nee record
// the vertex has written - the others sampled nothing that could contribute and have no record.
__device__ __forceinline__ bool resolve_add(const PathPool& pool, uint32_t slot, int nLights, bool usePhotons, f3& L) {
    const float4 pb = pool.pendBeta[slot];
    if (pb.w < 0.0f) return false;
    const uint32_t lights = (uint32_t)pb.w;
    const f3 beta = mk3(pb.x, pb.y, pb.z);
    f3 direct = mk3(0.0f);
    float maxVisibility = 0.0f;
    // (four lights' records are fetched together, then added in light order: one memory round trip for the usual scene instead
    // of one per light)
    for (int l0 = 0; l0 < nLights; l0 += 4) {
        float4 c4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int l = l0 + k;
            c4[k] = make_float4(0.0f, 0.0f, 0.0f, -1.0f);
            if (l < nLights && ((lights >> l) & 1u) != 0u) c4[k] = pool.nee[(size_t)l * pool.capacity + slot];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4 c = c4[k];
            if (c.w < 0.0f) continue;
            maxVisibility = fmaxf(maxVisibility, c.w);
            if (c.w <= 0.0f) continue;
            direct = direct + mk3(c.x, c.y, c.z) * c.w;
        }
    }
    if (max3c(direct) > 0.0f) L = L + beta * direct;
    if (usePhotons) {
        const float4 cz = pool.caustic[slot];
        if (cz.w > 0.0f) {
            const float causticScale = 0.1f + 0.9f * maxVisibility;
            L = L + beta * (mk3(cz.x, cz.y, cz.z) * causticScale);
        }
    }
    return true;
}
// the same, written back (tail kernel: the vertex is resolved as soon as its shadow chains are traced)
__device__ __forceinline__ void resolve_slot(const PathPool& pool, uint32_t slot, int nLights, bool usePhotons) {
    float4 L4 = pool.L[slot];
    f3 L = mk3(L4.x, L4.y, L4.z);
    if (!resolve_add(pool, slot, nLights, usePhotons, L)) return;
    pool.L[slot] = make_float4(L.x, L.y, L.z, L4.w);
    pool.pendBeta[slot] = make_float4(0.0f, 0.0f, 0.0f, -1.0f);
}

// One vertex of one path: returns true when the path continues (its next ray is in pool.rays[slot]).
// MICRO: the scene holds MtlMicrofacet materials (a second BSDF in the kernel; scenes without them run the leaner one).
// TAIL: the caller is tail_kernel, which walks a path to its end inside one kernel: the shadow segment of light l goes
// to the fixed entry slot * lights + l of pool.shadowQ instead of the shadow queue; the kernel traces it and resolves.
// PHOT: a caustics map may be present (pm.n decides at run time); false: the gather is not compiled in (it costs the
// shade kernel 8 ms per C3 frame just by being there).
// PHONG ("general materials"): the scene holds MtlPhong materials, or textures on Specular / Reflection / Refraction /
// Glossiness; false: only the Blinn form of the lobes and only the Diffuse / Emission / bump textures are compiled in (the
// Phong form costs the kernel 26 ms per C3 frame by being there, the four texture fetches 10 ms: every material of
// cerealBar.xml is MtlBlinn, textured in Diffuse, Emission and bump only).
// SPLIT: the lobe evaluation of the light loop is left to nee_kernel (kernels_nee.hip): the body samples the lights, appends
// the (vertex, light) pairs that can contribute to the NEE queue qNee and writes the vertex's record (nee_ctx.h) at neeIndex.
// ws: the slot the vertex's results are written to - `slot` itself, or a fresh slot of the compaction area when this bounce
// moves the paths together (pathstate.h "Compaction"; shade_kernel<..., REPACK>): everything the body reads is at `slot`,
// everything it leaves for later kernels (path state, next ray, NEE records and their slot word) at ws. The first-hit guides
// stay with the slot the path was generated in, where film_kernel looks for them.
// rayFlags: the path's PF_ flags, which travel in the spare word of its ray (pool.rays[2 * slot + 1].w) since round 3: the medium
// record is then read and written only by paths inside a medium, not by every vertex (32 bytes per vertex less).
template <bool MICRO, bool TAIL, bool PHOT = true, bool PHONG = true, bool SPLIT = false>
__device__ __forceinline__ bool shade_body(const DevScene& sc, const DevCamera& cam, const PathPool& pool, const DevPhotonMap& pmArg,
                                           uint32_t slot, const RayT& ray, uint32_t rayFlags, const HitT& h, int bounce, int maxBounces,
                                           unsigned long long seed, int sBase, int S, const int* __restrict__ pixels,
                                           const SegQ& qShadow, uint32_t neeIndex = 0u, const SegQ* qNee = nullptr,
                                           const DevLight* lightsArg = nullptr, uint32_t ws = 0xFFFFFFFFu, bool* culledOut = nullptr) {
    if (ws == 0xFFFFFFFFu) ws = slot;
    // the lights, read where the kernel's argument holds them: shade_kernel shades with a copy of the scene descriptor whose
    // table pointers lead to LDS, and an array indexed at run time would keep that whole copy in scratch memory
    const DevLight* lights = lightsArg ? lightsArg : sc.lights;
    DevPhotonMap pm = pmArg;
    if (!PHOT) pm.n = 0;   // every `pm.n > 0` below folds to false
    bool alive = false;
    const uint32_t cap = pool.capacity;
    const uint32_t shadowMult = (uint32_t)max(1, sc.n_lights);
    {
        // a camera ray carries the initial path state (raygen_kernel does not write it out: 48 B per path saved each way)
        float4 L4 = make_float4(0.0f, 0.0f, 0.0f, RTX_BIG);
        float4 b4 = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
        float4 m4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        uint32_t flags = PF_LAST_DELTA;
        if (bounce > 0) {
            L4 = pool.L[slot];
            b4 = pool.beta[slot];
            flags = rayFlags;
            if (flags & PF_INSIDE_MEDIUM) m4 = pool.medium[slot];
        }
        f3 L = mk3(L4.x, L4.y, L4.z);
#ifndef RTX_ABLATE_RESOLVE   // (RTX_ABLATE_*: timing-only builds of tools/build_variant.sh that leave a part of the body out; wrong images)
        if (bounce > 0) resolve_add(pool, slot, sc.n_lights, pm.n > 0, L);   // the NEE of the previous vertex (its shadow results are in)
#endif
        f3 beta = mk3(b4.x, b4.y, b4.z);
        float lastPdf = b4.w;
        f3 mediumAbs = mk3(m4.x, m4.y, m4.z);
        const f3 rdir = mk3(ray.dx, ray.dy, ray.dz), rorg = mk3(ray.ox, ray.oy, ray.oz);
        // pixel / sample of this path -> RNG key
        const uint32_t oslot = path_origin(pool, slot);
        const uint32_t lane64 = oslot & 63u, cs = oslot >> 6;
        const int pix = pixels[(cs / (uint32_t)S) * 64u + lane64];
        const int sIdx = sBase + (int)(cs % (uint32_t)S);
        bool pending = false;   // NEE of this bounce to be resolved
        bool culled = false;    // the vertex is left to the general kernels (below): nothing is written
        uint32_t pendMask = sc.n_lights >= 32 ? 0xFFFFFFFFu : (1u << sc.n_lights) - 1u;   // lights whose nee records this vertex writes (resolve_add)

        do {
            if (h.slot == -1) {
                // ---- miss: backplate / environment (RayCaster.cpp:362-382) ----
#ifdef RTX_ABLATE_MISS
                L = L + beta * 0.5f;
                break;
#endif
                const f3 nd = normalized3(rdir);
                float uEnv = 0.0f, vEnv = 0.0f;
                dir_to_screen_uv(cam, nd, uEnv, vEnv);
                const bool hasBackground = has_map_or_value(sc.bg, sc.bg_tex);
                const bool hasEnvironment = has_map_or_value(sc.env, sc.env_tex);
                const f3 backplate = hasBackground ? eval_tc(sc, sc.bg, sc.bg_tex, mk3(uEnv, vEnv, 0.0f)) : mk3(0.0f);
                f3 envLight = mk3(0.0f);
                if (hasEnvironment) envLight = sc.env_tex >= 0 ? ld3(sc.env) * tex_eval_env(sc, sc.env_tex, nd) : ld3(sc.env);
                const f3 bgEmit = (hasBackground && sc.bg_emission > 0.0f) ? backplate * sc.bg_emission : mk3(0.0f);
                if (bounce == 0) {
                    const f3 vis = hasBackground ? backplate : envLight;
                    L = L + beta * (vis + bgEmit);
                    pool.albedo[oslot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);   // no first hit: no guides (raygen leaves them to this kernel)
                    pool.normal[oslot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                } else {
                    L = L + beta * (envLight + bgEmit);
                }
                break;
            }
            SurfHit sh;
            if (h.slot >= 0) resolve_triangle(sc, ray, h, sh);
            else resolve_light(lights, ray, h, sh);
            // Lean kernels (shade_kernel of a scene of world-space triangles): a hit that the reference's box tests would have
            // culled - not checked by the lean extend kernel - leaves the vertex alone; its path goes to the general tail kernel,
            // which traces the ray again (rtx_render.cpp). Here, where the triangle record is at hand: a triangle well inside
            // both boxes (TRI_INTERIOR, most of them) costs the check no memory access at all.
            if (!TAIL && culledOut && pool.cullPaths && h.slot >= 0 && !ref_reaches_interior(sh.triWord, ray, h.t)) {
                float ob[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) ob[i] = sc.objXf[sh.obj].box[i];
                if (!ref_reaches_box<false>(sc, ray, ob, h.t)) {
                    const uint32_t k = atomicAdd(&pool.totals[4], 1u);
                    if (k < RTX_CULL_CAP) {
                        pool.cullPaths[k] = slot;
                        culled = true;
                        break;
                    }
                }
            }
            const float hitDist = len3(sh.p - rorg);
            if ((flags & PF_INSIDE_MEDIUM) && hitDist > 0.0f) beta = apply_absorption(beta, mediumAbs, hitDist);

            const DevMaterial* mtl = nullptr;
            int mtlIndex = -1;
            if (sh.obj >= 0) {
                mtlIndex = sc.objs[sh.obj].material;
                if (mtlIndex >= 0) mtl = &sc.mtls[mtlIndex];
            }
#ifndef RTX_ABLATE_BUMP
            if (mtl) apply_bump(sc, *mtl, sh);
#endif
            SampCtx sx;
            sx.P = sh.p;
            sx.V = -normalized3(rdir);
            sx.N = sh.N;
            sx.GN = sh.GN;
            sx.uvw = sh.uvw;
            sx.front = sh.front;
            sx.key = rng_key(seed, (uint32_t)pix, (uint32_t)sIdx);
            sx.stream = (uint32_t)bounce;

            const bool micro = MICRO && mtl && mtl->type == RTX_MTL_MICROFACET;
            if (bounce == 0) {   // :401-407, GetDiffuseAlbedo :135-138
                L4.w = hitDist;
                const f3 alb = (mtl && !micro) ? eval_tc(sc, mtl->kd, mtl->kd_tex, sx.uvw) : mk3(0.5f);
                const f3 fn = safe_normalize3(sh.GN);
                pool.albedo[oslot] = make_float4(alb.x, alb.y, alb.z, 1.0f);
                pool.normal[oslot] = make_float4(fn.x, fn.y, fn.z, 1.0f);
            }

            if (sh.light >= 0) {   // :410-439
                const DevLight& light = lights[sh.light];
                const f3 emitted = light.size > 0.0f ? ld3(light.radiance) : ld3(light.intensity) / (PI_F * light.size * light.size);
                float lightPdfSA = 0.0f;
                if (flags & PF_HAS_PREV) {
                    const float4 pp = pool.prevP[slot];
                    DirInfo li;
                    light_get_sample_info(light, mk3(pp.x, pp.y, pp.z), rdir, li);
                    lightPdfSA = li.prob;
                    if (light.size > 0.0f) {
                        const float cosLight = fmaxf(0.0f, dot3(-sh.N, safe_normalize3(rdir)));
                        const float dist2 = li.dist * li.dist;
                        if (cosLight > 1e-6f && dist2 > 0.0f) lightPdfSA = li.prob * dist2 / cosLight;
                        else lightPdfSA = 0.0f;
                    } else
                        lightPdfSA = 0.0f;
                }
                const float w = ((flags & PF_LAST_DELTA) || lightPdfSA <= 0.0f) ? 1.0f : power_heuristic(lastPdf, lightPdfSA);
                L = L + beta * emitted * w;
                break;
            }
            if (!mtl) break;

            const f3 emission = micro ? mk3(0.0f) : eval_tc(sc, mtl->ke, mtl->ke_tex, sx.uvw);   // :445-446, GetEmission :158-161
            if (max3c(emission) > 0.0f) L = L + beta * emission;

            PhongCtx ctx;
            MicroCtx mctx;
            if (micro) mctx = build_micro_ctx(sc, *mtl, sx);
            else ctx = build_phong_ctx<PHONG>(sc, *mtl, sx);
            const bool phong = PHONG && mtl->type == RTX_MTL_PHONG;

            // ---- next event estimation: contributions now, visibility by the shadow kernel (:456-503) ----
            const bool photonSurface = mtl_is_photon_surface(*mtl);
            const bool wantCaustic = pm.n > 0 && photonSurface;
            if (SPLIT) {
                // Stage A of the light loop, for every light at once: the point on the light and the rejections that need no
                // lobe evaluation (no sample; light behind a surface that lets none through and no caustics map asking for the
                // visibility: GetSampleInfo returns nothing there, Shading.cpp:639-646, MtlMicrofacet :905). The pairs that are
                // left - one in four - go to the NEE queue with their direction; nee_kernel evaluates them packed. The vertex's
                // record is written only if a pair is left, and pendBeta.w names the lights whose nee records exist.
                const f3 cN = micro ? mctx.N : ctx.N;
                const bool behindToo = wantCaustic || (micro ? mctx.pt > 0.0f : luma1(ctx.transColor) > 0.0f);
                // what every pair of this vertex carries in its queue entry: the flags word and the context's alpha (nee_ctx.h)
                const bool compactRec = !micro && mtlIndex < 65536 && phong_inputs_constant<PHONG>(*mtl);
                uint32_t nf = (sx.front ? NEEF_FRONT : 0u) | (phong ? NEEF_PHONG : 0u) | (wantCaustic ? NEEF_CAUSTIC : 0u);
                nf = micro ? nee_flags_micro(nf, mctx) : nee_flags_phong(nf, ctx);
                if (compactRec) nf |= NEEF_COMPACT | ((uint32_t)mtlIndex << 16);
                const float recAlpha = micro ? mctx.alpha : ctx.alpha;
                uint32_t lightMask = 0u;
#ifdef RTX_ABLATE_NEE
                for (int l0 = 0; l0 < 0; l0 += 4) {
#else
                for (int l0 = 0; l0 < sc.n_lights; l0 += 4) {
#endif
                    f3 ld[4];
                    float dd[4];
                    unsigned long long bal[4];
                    uint32_t waveTotal = 0u;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int l = l0 + k;
                        bool surv = false;
                        ld[k] = mk3(0.0f);
                        dd[k] = 0.0f;
                        if (l < sc.n_lights) {
                            const DevLight& light = lights[l];
                            const float u1 = sx.rnd(DIM_LIGHT0 + 2 * l), u2 = sx.rnd(DIM_LIGHT0 + 2 * l + 1);
                            if (light_sample_dir(light, sx.P, u1, u2, ld[k], dd[k])) {
                                DirInfo li;
                                light_sample_terms(light, dd[k], li);
                                if (li.prob > 0.0f && max3c(li.mult) > 0.0f) {
                                    // the sign of N . dir, where dir is the sampled direction normalised twice more on its way into
                                    // GetSampleInfo (RayCaster.cpp:475, Shading.cpp:535). ld[k] is a unit vector already: the two
                                    // normalisations move the product by a few 1e-7 at most, so away from zero its own sign decides
                                    const float d0 = dot3(cN, ld[k]);
                                    surv = behindToo || d0 > 1e-5f || (d0 >= -1e-5f && dot3(cN, safe_normalize3(safe_normalize3(ld[k]))) > 0.0f);
                                }
                            }
                        }
                        if (surv) lightMask |= 1u << l;
                        bal[k] = __ballot(surv);
                        waveTotal += (uint32_t)__popcll(bal[k]);
                    }
                    if (waveTotal) {   // one atomic per wave for the group's pairs; entries light by light
                        uint32_t at = segq_reserve(*qNee, waveTotal, (uint32_t)sc.n_lights);
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            if ((lightMask >> (l0 + k)) & 1u) {
                                const size_t e = at + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal[k] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal[k], 0u));
                                pool.neeQ[2 * e] = make_float4(ld[k].x, ld[k].y, ld[k].z, dd[k]);
                                pool.neeQ[2 * e + 1] = make_float4(__uint_as_float(neeIndex), __uint_as_float(ws), __uint_as_float(nf | ((uint32_t)(l0 + k) << 8)), recAlpha);
                            }
                            at += (uint32_t)__popcll(bal[k]);
                        }
                    }
                }
                if (lightMask) {
                    float4* rec = pool.neeCtx + neeIndex;
                    if (micro) nee_ctx_store_micro(rec, pool.neeStride, sh.p, sx.GN, mctx);
                    else if (compactRec) nee_ctx_store_compact(rec, pool.neeStride, sh.p, sx.GN, ctx);
                    else nee_ctx_store(rec, pool.neeStride, sh.p, sx.GN, ctx);
                    pending = true;
                    pendMask = lightMask;
                }
            } else {
                for (int l = 0; l < sc.n_lights; ++l) {
                    const DevLight& light = lights[l];
                    f3 ldir;
                    DirInfo li;
                    float4 contrib = make_float4(0.0f, 0.0f, 0.0f, -1.0f);
                    const float u1 = sx.rnd(DIM_LIGHT0 + 2 * l), u2 = sx.rnd(DIM_LIGHT0 + 2 * l + 1);
                    if (light_generate_sample(light, sx.P, u1, u2, ldir, li) && li.prob > 0.0f && max3c(li.mult) > 0.0f) {
                        const float shadowMax = (li.dist > 0.0f && li.dist < RTX_BIG * 0.5f) ? li.dist : RTX_BIG;
                        DirInfo bs;
                        if (micro) evaluate_microfacet(mctx, sx, safe_normalize3(ldir), bs);
                        else evaluate_phong_blinn(ctx, phong, sx, safe_normalize3(ldir), bs);
                        f3 c = mk3(0.0f);
                        if (bs.prob > 0.0f && max3c(bs.mult) > 0.0f) {
                            if (light.size > 0.0f) {
                                const float dist = li.dist;
                                const float cosLight = light_cos_for_nee(light, sh.p, safe_normalize3(ldir));
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
                        // the shadow segment is needed when it can contribute, or when its visibility scales the
                        // caustic term (:468, :518)
                        if (max3c(c) > 0.0f || wantCaustic) {
                            contrib = make_float4(c.x, c.y, c.z, 0.0f);
                            pending = true;
                            const f3 so = offset_point(sh.p, sx.GN, ldir);
                            const f3 sd = safe_normalize3(ldir);
                            // wave-wide append of the lanes that reach this point (ballot counts active lanes only)
                            // (tail_kernel: a fixed entry per path and light, by the slot the path was generated in - the queue arrays
                            // are laid out for the slots of the pass)
                            const uint32_t qs = TAIL ? oslot * shadowMult + (uint32_t)l : segq_append(qShadow, true, shadowMult);
                            pool.shadowQ[2 * (size_t)qs] = make_float4(so.x, so.y, so.z, shadowMax);
                            pool.shadowQ[2 * (size_t)qs + 1] = make_float4(sd.x, sd.y, sd.z, __uint_as_float(ws * 8u + (uint32_t)l));
                        }
                    }
                    pool.nee[(size_t)l * cap + ws] = contrib;
                }
            }
            if (pending) pool.pendBeta[ws] = make_float4(beta.x, beta.y, beta.z, (float)pendMask);

            if (pm.n > 0) {   // :508-522 (scaled by the visibility in resolve_add)
                float4 cz = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (photonSurface) {
                    f3 irrad = estimate_irradiance(pm, sh.p, sh.N);
                    if (max3c(irrad) > 0.0f) {
                        const f3 kd = micro ? mk3(0.5f) : eval_tc(sc, mtl->kd, mtl->kd_tex, sx.uvw);
                        irrad = irrad * 0.1f;   // kCausticBoost
                        const f3 t = kd * (1.0f / PI_F) * irrad;
                        cz = make_float4(t.x, t.y, t.z, 1.0f);
                        if (!pending) pool.pendBeta[ws] = make_float4(beta.x, beta.y, beta.z, 0.0f);
                        pending = true;
                    }
                }
                pool.caustic[ws] = cz;
            }

            // ---- BSDF sampling for the next bounce (:525-566) ----
            f3 nextDir;
            DirInfo bsd;
            if (micro ? !sample_microfacet(mctx, sx, nextDir, bsd) : !sample_phong_blinn(ctx, phong, sx, nextDir, bsd)) break;
            if (bsd.prob <= 0.0f || max3c(bsd.mult) <= 0.0f) break;
            const float pdfThroughput = fmaxf(1e-4f, bsd.prob);
            beta = beta * (bsd.mult / pdfThroughput);
            lastPdf = bsd.prob;
            flags = (flags & ~PF_LAST_DELTA) | (bsd.dist < 0.0f ? PF_LAST_DELTA : 0u);
            pool.prevP[ws] = make_float4(sh.p.x, sh.p.y, sh.p.z, 0.0f);
            flags |= PF_HAS_PREV;
            if (bsd.lobe & LOBE_TRANSMISSION) {
                if (sx.front) {
                    flags |= PF_INSIDE_MEDIUM;
                    mediumAbs = ld3(mtl->absorption);
                } else {
                    flags &= ~PF_INSIDE_MEDIUM;
                    mediumAbs = mk3(0.0f);
                }
            }
            if (bounce >= 5) {
                float q = clamp01f(max3c(beta));
                q = fmaxf(q, 0.05f);
                if (sx.rnd(DIM_RR) > q) break;
                beta = beta * (1.0f / q);
            }
            if (bounce + 1 >= maxBounces) break;
            const f3 no = offset_point(sh.p, sx.GN, nextDir);
            const f3 nd = safe_normalize3(nextDir);
            pool.rays[2 * (size_t)ws] = make_float4(no.x, no.y, no.z, RTX_BIG);
            pool.rays[2 * (size_t)ws + 1] = make_float4(nd.x, nd.y, nd.z, __uint_as_float(flags));
            alive = true;
        } while (false);

        if (culled) {
            *culledOut = true;
            return false;
        }
        pool.L[ws] = make_float4(L.x, L.y, L.z, L4.w);
        if (alive) {   // (a path that has ended leaves its radiance and what is pending; nobody reads the rest)
            pool.beta[ws] = make_float4(beta.x, beta.y, beta.z, lastPdf);
            if (flags & PF_INSIDE_MEDIUM) pool.medium[ws] = make_float4(mediumAbs.x, mediumAbs.y, mediumAbs.z, 0.0f);
        }
        if (!pending) {
            // mark "nothing to resolve" for resolve_add
            pool.pendBeta[ws] = make_float4(0.0f, 0.0f, 0.0f, -1.0f);
        }
    }
    return alive;
}
