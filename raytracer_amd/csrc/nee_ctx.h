// The record shade_kernel leaves behind for nee_kernel: what the lobe evaluation of the light loop of TracePath (reference
// RayTracer/RayCaster.cpp:456-503) needs of one path vertex - the hit point, the two normals, the view direction and the
// lobe terms BuildPhongContext has computed (Shading.cpp:227-278; the reference rebuilds them for every GetSampleInfo
// call, :721) - so that the light samples that can contribute are evaluated one lane per (vertex, light) in a kernel of
// their own (kernels_nee.hip). 128 bytes = 8 x float4 per vertex, indexed by the vertex's position i in the bounce's queue
// and stored word by word, w[k] at neeCtx[k * stride + i]: the 64 lanes of a wave then write 1 KB of consecutive memory per
// store (as one record per lane, every store touched 64 different lines with 16 bytes each: 512 write requests per wave
// instead of 128). Written only for vertices that have a pair in the NEE queue.
//
//   w0  P.xyz            | path slot
//   w1  -, -             | flags | alpha
//   w2  N.xyz            | ior            (N: shading normal, flipped to the side of the ray as the context holds it)
//   w3  GN.xyz           | kd.x
//   w4  V.xyz            | kd.y
//   w5  specColor.xyz    | kd.z           microfacet: F0
//   w6  kr + fresnelRef  | -              microfacet: diffuse.xyz | pd
//   w7  transColor.xyz   | -              microfacet: trans.xyz   | ps           (kd.xyz = pt, total, -)
#pragma once
#include "bsdf.h"

#define NEE_CTX_VEC 8u            // float4 per record
#define NEEF_FRONT 1u
#define NEEF_PHONG 2u             // MtlPhong (else MtlBlinn)
#define NEEF_MICRO 4u             // MtlMicrofacet: the record holds a MicroCtx
#define NEEF_CAUSTIC 8u           // a caustics map scales with this vertex's visibility: every light sample needs its shadow ray (RayCaster.cpp:518)
#define NEEF_KRPOS 16u            // max(kr) > 0 (Shading.cpp:640)
#define NEEF_PTPOS 32u            // luma(transColor) > 0: light from behind the surface can contribute

#define NEE_W(rec, k) (rec)[(size_t)(k) * stride]
__device__ __forceinline__ void nee_ctx_store(float4* rec, size_t stride, uint32_t slot, uint32_t flags, f3 P, f3 GN,
                                              const PhongCtx& c) {
    if (c.krPos) flags |= NEEF_KRPOS;
    if (luma1(c.transColor) > 0.0f) flags |= NEEF_PTPOS;
    NEE_W(rec, 0) = make_float4(P.x, P.y, P.z, __uint_as_float(slot));
    NEE_W(rec, 1) = make_float4(0.0f, 0.0f, __uint_as_float(flags), c.alpha);
    NEE_W(rec, 2) = make_float4(c.N.x, c.N.y, c.N.z, c.ior);
    NEE_W(rec, 3) = make_float4(GN.x, GN.y, GN.z, c.kd.x);
    NEE_W(rec, 4) = make_float4(c.V.x, c.V.y, c.V.z, c.kd.y);
    NEE_W(rec, 5) = make_float4(c.specColor.x, c.specColor.y, c.specColor.z, c.kd.z);
    NEE_W(rec, 6) = make_float4(c.krf.x, c.krf.y, c.krf.z, 0.0f);
    NEE_W(rec, 7) = make_float4(c.transColor.x, c.transColor.y, c.transColor.z, 0.0f);
}
__device__ __forceinline__ void nee_ctx_store_micro(float4* rec, size_t stride, uint32_t slot, uint32_t flags, f3 P, f3 GN,
                                                    const MicroCtx& c) {
    flags |= NEEF_MICRO;
    if (c.pt > 0.0f) flags |= NEEF_PTPOS;
    NEE_W(rec, 0) = make_float4(P.x, P.y, P.z, __uint_as_float(slot));
    NEE_W(rec, 1) = make_float4(0.0f, 0.0f, __uint_as_float(flags), c.alpha);
    NEE_W(rec, 2) = make_float4(c.N.x, c.N.y, c.N.z, c.ior);
    NEE_W(rec, 3) = make_float4(GN.x, GN.y, GN.z, c.pt);
    NEE_W(rec, 4) = make_float4(c.V.x, c.V.y, c.V.z, c.total);
    NEE_W(rec, 5) = make_float4(c.F0.x, c.F0.y, c.F0.z, 0.0f);
    NEE_W(rec, 6) = make_float4(c.diffuse.x, c.diffuse.y, c.diffuse.z, c.pd);
    NEE_W(rec, 7) = make_float4(c.trans.x, c.trans.y, c.trans.z, c.ps);
}
// the evaluation side: words 2..7 of a record (w1 is read by the caller: flags, alpha)
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
