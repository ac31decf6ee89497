// The record shade_kernel leaves behind for nee_kernel: what the lobe evaluation of the light loop of TracePath (reference
// RayTracer/RayCaster.cpp:456-503) needs of one path vertex - the hit point, the two normals, the view direction and the
// lobe terms BuildPhongContext has computed (Shading.cpp:227-278; the reference rebuilds them for every GetSampleInfo
// call, :721) - so that the light samples that can contribute are evaluated one lane per (vertex, light) in a kernel of
// their own (kernels_nee.hip). 128 bytes = 8 x float4 per vertex, indexed by the vertex's position i in the bounce's queue
// and stored word by word, w[k] at neeCtx[k * stride + i]: the 64 lanes of a wave then write 1 KB of consecutive memory per
// store (as one record per lane, every store touched 64 different lines with 16 bytes each: 512 write requests per wave
// instead of 128). Written only for vertices that have a pair in the NEE queue.
//
//   w0  P.xyz            | -              (the path slot, the flags and alpha travel in the pair's queue entry since round 3:
//   w1  -                                  neeQ[2e+1] = {vertex index, path slot, flags | light << 8 | material << 16, alpha})
//   w2  N.xyz            | ior            (N: shading normal, flipped to the side of the ray as the context holds it)
//   w3  GN.xyz           | kd.x
//   w4  V.xyz            | kd.y
//   w5  specColor.xyz    | kd.z           microfacet: F0
//   w6  kr + fresnelRef  | -              microfacet: diffuse.xyz | pd
//   w7  transColor.xyz   | -              microfacet: trans.xyz   | ps           (kd.xyz = pt, total, -)
//
// Compact form (NEEF_COMPACT, round 3): a Blinn / Phong vertex whose material puts no texture on the context's inputs leaves
// 48 bytes - w0, w2, w3, the view direction in their spare words, and the material's index in the upper half of the flags word -
// and nee_kernel forms kd / specColor / krf / transColor / alpha / ior itself, from the material record, with the arithmetic
// BuildPhongContext uses (bsdf.h finish_phong_ctx): the same operations on the same operands. nee_kernel moves bytes - 128 of
// them per pair for this record before - and half a record less took 20 ms off a C3 frame.
//   w0  P.xyz            | V.x
//   w2  N.xyz            | V.y
//   w3  GN.xyz           | V.z
#pragma once
#include "bsdf.h"

#define NEE_CTX_VEC 8u            // float4 per record
#define NEEF_FRONT 1u
#define NEEF_PHONG 2u             // MtlPhong (else MtlBlinn)
#define NEEF_MICRO 4u             // MtlMicrofacet: the record holds a MicroCtx
#define NEEF_CAUSTIC 8u           // a caustics map scales with this vertex's visibility: every light sample needs its shadow ray (RayCaster.cpp:518)
#define NEEF_KRPOS 16u            // max(kr) > 0 (Shading.cpp:640)
#define NEEF_PTPOS 32u            // luma(transColor) > 0: light from behind the surface can contribute
#define NEEF_COMPACT 64u          // compact record: the lobe terms are rebuilt from material (flags >> 16)

#define NEE_W(rec, k) (rec)[(size_t)(k) * stride]
// the flags word of a vertex's pairs (neeQ[2e+1].z; .w = the context's alpha)
__device__ __forceinline__ uint32_t nee_flags_phong(uint32_t flags, const PhongCtx& c) {
    if (c.krPos) flags |= NEEF_KRPOS;
    if (luma1(c.transColor) > 0.0f) flags |= NEEF_PTPOS;
    return flags;
}
__device__ __forceinline__ uint32_t nee_flags_micro(uint32_t flags, const MicroCtx& c) {
    flags |= NEEF_MICRO;
    if (c.pt > 0.0f) flags |= NEEF_PTPOS;
    return flags;
}
__device__ __forceinline__ void nee_ctx_store(float4* rec, size_t stride, f3 P, f3 GN, const PhongCtx& c) {
    NEE_W(rec, 0) = make_float4(P.x, P.y, P.z, 0.0f);
    NEE_W(rec, 2) = make_float4(c.N.x, c.N.y, c.N.z, c.ior);
    NEE_W(rec, 3) = make_float4(GN.x, GN.y, GN.z, c.kd.x);
    NEE_W(rec, 4) = make_float4(c.V.x, c.V.y, c.V.z, c.kd.y);
    NEE_W(rec, 5) = make_float4(c.specColor.x, c.specColor.y, c.specColor.z, c.kd.z);
    NEE_W(rec, 6) = make_float4(c.krf.x, c.krf.y, c.krf.z, 0.0f);
    NEE_W(rec, 7) = make_float4(c.transColor.x, c.transColor.y, c.transColor.z, 0.0f);
}
__device__ __forceinline__ void nee_ctx_store_compact(float4* rec, size_t stride, f3 P, f3 GN, const PhongCtx& c) {
    NEE_W(rec, 0) = make_float4(P.x, P.y, P.z, c.V.x);
    NEE_W(rec, 2) = make_float4(c.N.x, c.N.y, c.N.z, c.V.y);
    NEE_W(rec, 3) = make_float4(GN.x, GN.y, GN.z, c.V.z);
}
// ... and its evaluation side: the context BuildPhongContext would build at this vertex (w0, w2, w3: the record's words)
__device__ __forceinline__ PhongCtx nee_ctx_rebuild(const DevMaterial& m, float4 w0, float4 w2, float4 w3, bool front, bool phong) {
    PhongCtx c;
    c.N = mk3(w2.x, w2.y, w2.z);
    c.V = mk3(w0.w, w2.w, w3.w);
    c.R = phong ? safe_normalize3(reflect3(-c.V, c.N)) : mk3(0.0f);
    c.kd = ld3(m.kd);          // TexturedColor::Eval without a map: the value
    c.ks = ld3(m.ks);
    c.kr = ld3(m.kr);
    c.kt = ld3(m.kt);
    c.alpha = fmaxf(0.001f, m.gloss);
    c.ior = m.ior;
    finish_phong_ctx(c, front);
    return c;
}
__device__ __forceinline__ void nee_ctx_store_micro(float4* rec, size_t stride, f3 P, f3 GN, const MicroCtx& c) {
    NEE_W(rec, 0) = make_float4(P.x, P.y, P.z, 0.0f);
    NEE_W(rec, 2) = make_float4(c.N.x, c.N.y, c.N.z, c.ior);
    NEE_W(rec, 3) = make_float4(GN.x, GN.y, GN.z, c.pt);
    NEE_W(rec, 4) = make_float4(c.V.x, c.V.y, c.V.z, c.total);
    NEE_W(rec, 5) = make_float4(c.F0.x, c.F0.y, c.F0.z, 0.0f);
    NEE_W(rec, 6) = make_float4(c.diffuse.x, c.diffuse.y, c.diffuse.z, c.pd);
    NEE_W(rec, 7) = make_float4(c.trans.x, c.trans.y, c.trans.z, c.ps);
}
// the evaluation side: words 2..7 of a record (flags, alpha: the pair's queue entry)
__device__ __forceinline__ PhongCtx nee_ctx_load(const float4* rec, size_t stride, float alpha, uint32_t flags, bool phong) {
    const float4 w2 = NEE_W(rec, 2), w3 = NEE_W(rec, 3), w4 = NEE_W(rec, 4), w5 = NEE_W(rec, 5), w6 = NEE_W(rec, 6), w7 = NEE_W(rec, 7);
    PhongCtx c;
    c.N = mk3(w2.x, w2.y, w2.z);
    c.ior = w2.w;
    c.V = mk3(w4.x, w4.y, w4.z);
    c.kd = mk3(w3.w, w4.w, w5.w);
    c.specColor = mk3(w5.x, w5.y, w5.z);
    c.krf = mk3(w6.x, w6.y, w6.z);
    c.transColor = mk3(w7.x, w7.y, w7.z);
    c.alpha = alpha;
    c.krPos = (flags & NEEF_KRPOS) != 0u;
    c.R = phong ? safe_normalize3(reflect3(-c.V, c.N)) : mk3(0.0f);   // as BuildPhongContext forms it (Shading.cpp:233)
    c.ks = c.kr = c.kt = c.fresnelRef = mk3(0.0f);                     // (not read by the evaluation)
    return c;
}
__device__ __forceinline__ MicroCtx nee_ctx_load_micro(const float4* rec, size_t stride, float alpha) {
    const float4 w2 = NEE_W(rec, 2), w3 = NEE_W(rec, 3), w4 = NEE_W(rec, 4), w5 = NEE_W(rec, 5), w6 = NEE_W(rec, 6), w7 = NEE_W(rec, 7);
    MicroCtx c;
    c.N = mk3(w2.x, w2.y, w2.z);
    c.ior = w2.w;
    c.V = mk3(w4.x, w4.y, w4.z);
    c.pt = w3.w;
    c.total = w4.w;
    c.F0 = mk3(w5.x, w5.y, w5.z);
    c.diffuse = mk3(w6.x, w6.y, w6.z);
    c.pd = w6.w;
    c.trans = mk3(w7.x, w7.y, w7.z);
    c.ps = w7.w;
    c.alpha = alpha;
    return c;
}
