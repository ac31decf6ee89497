// Device-side BSDF and light sampling: MtlBlinn/MtlPhong::GenerateSample/GetSampleInfo and
// PointLight::GenerateSample/GetSampleInfo/RandomPhoton of the reference (RayTracer/Shading.cpp:31-737),
// written for per-lane evaluation inside the wavefront shade kernel. Random numbers come from the counter-based
// generator of vec.h keyed by (pixel, sample | photon) x bounce x draw slot.
#pragma once
#include "surface.h"

#define LOBE_NONE 0
#define LOBE_DIFFUSE 1
#define LOBE_SPECULAR 2
#define LOBE_TRANSMISSION 4
#define LOBE_ALL 7
#define PI_F 3.14159265358979323846f

struct DirInfo {   // DirSampler::Info
    f3 mult;
    float prob, dist;
    int lobe;
};
__device__ __forceinline__ void set_void(DirInfo& s) {
    s.mult = mk3(0.0f);
    s.prob = 0.0f;
    s.dist = 0.0f;
    s.lobe = LOBE_NONE;
}

// SamplerInfo after SetHit (RayCaster.cpp:393-399) + RNG key
struct SampCtx {
    f3 P, V, N, GN, uvw;
    bool front;
    unsigned long long key;
    uint32_t stream;
    __device__ __forceinline__ float rnd(uint32_t dim) const { return rng_draw(key, stream, dim); }
};

__device__ __forceinline__ void build_onb(f3 n, f3& u, f3& v) {   // Shading.cpp:31-37
    if (fabsf(n.x) > 0.5f) u = cross3(mk3(0.0f, 1.0f, 0.0f), n);
    else u = cross3(mk3(1.0f, 0.0f, 0.0f), n);
    u = normalized3(u);
    v = cross3(n, u);
    v = normalized3(v);
}
__device__ __forceinline__ f3 reflect3(f3 w, f3 n) { return w - 2.0f * dot3(w, n) * n; }   // :39-43
__device__ __forceinline__ float transmission_scale(float eta) { return fminf(eta * eta, 10.0f); }   // :45-48
__device__ __forceinline__ float fresnel_dielectric(float cosThetaI, float etaI, float etaT) {        // :50-61
    cosThetaI = clamp01f(fabsf(cosThetaI));
    const float eta = etaI / etaT;
    const float sin2T = eta * eta * (1.0f - cosThetaI * cosThetaI);
    if (sin2T >= 1.0f) return 1.0f;
    const float cosThetaT = sqrtf(fmaxf(0.0f, 1.0f - sin2T));
    const float rParl = ((etaT * cosThetaI) - (etaI * cosThetaT)) / ((etaT * cosThetaI) + (etaI * cosThetaT));
    const float rPerp = ((etaI * cosThetaI) - (etaT * cosThetaT)) / ((etaI * cosThetaI) + (etaT * cosThetaT));
    return clamp01f(0.5f * (rParl * rParl + rPerp * rPerp));
}

struct PhongCtx {
    f3 N, V, R, kd, ks, kr, kt;
    float alpha, ior;
    // lobe terms shared by sampling and evaluation (Shading.cpp:282-300 == :535-546)
    f3 fresnelRef, transColor, specColor;
    f3 krf;       // kr + fresnelRef: the colour of the reflection lobe before its clamp (:290, :541)
    bool krPos;   // max(kr) > 0 (:640)
};

// Shading.cpp:227-278 — built ONCE per hit (the reference rebuilds it for every GenerateSample/GetSampleInfo call)
// SPECTEX = false: no material of the scene puts a texture on Specular / Reflection / Refraction / Glossiness (a scene fact,
// rtx_scene::generalMtl): the four inlined texture fetches are then not compiled in
// The lobe terms of a context from its raw inputs - c.kd / ks / kr / kt / alpha / ior as Diffuse() ... Glossiness() evaluate, c.N and
// c.V set - by the arithmetic of Shading.cpp:241-278. Apart from build_phong_ctx, nee_kernel runs it to rebuild the terms of a
// vertex whose material holds no texture from the material record (nee_ctx.h, compact records): the same operations on the same
// operands, hence the same bits.
__device__ __forceinline__ void finish_phong_ctx(PhongCtx& c, bool front) {
    c.kd = clamp_max3(c.kd, 1.0f);
    c.ks = clamp_max3(c.ks, 1.0f);
    c.kr = clamp_max3(c.kr, 1.0f);
    c.kt = clamp_max3(c.kt, 1.0f);
    const float krMax = max3c(c.kr);
    const float ktMax = max3c(c.kt);
    if (krMax <= 0.0f && ktMax <= 0.0f) c.ks = clamp_max3(c.ks, 0.08f);
    const float reserve = fminf(1.0f, krMax + ktMax);
    const float available = fmaxf(0.0f, 1.0f - reserve);
    const f3 kdks = c.kd + c.ks;
    const float kdksMax = max3c(kdks);
    if (kdksMax > 0.0f && kdksMax > available) {
        const float scale = available / kdksMax;
        c.kd = c.kd * scale;
        c.ks = c.ks * scale;
    }
    const float etaI = front ? 1.0f : c.ior;
    const float etaT = front ? c.ior : 1.0f;
    const float F = (c.ior > 0.0f && max3c(c.kt) > 0.0f) ? fresnel_dielectric(fabsf(dot3(c.N, c.V)), etaI, etaT) : 0.0f;
    c.fresnelRef = c.kt * F;
    c.transColor = c.kt * (1.0f - F);
    c.specColor = c.ks;
    if (max3c(c.kr) > 0.0f) c.specColor = mk3(0.0f);
    c.specColor = clamp_max3(c.specColor, 1.0f);
    c.krf = c.kr + c.fresnelRef;
    c.krPos = max3c(c.kr) > 0.0f;
}
template <bool SPECTEX = true>
__device__ __forceinline__ PhongCtx build_phong_ctx(const DevScene& sc, const DevMaterial& m, const SampCtx& s) {
    PhongCtx c;
    c.N = normalized3(s.N);
    if (!s.front) c.N = -c.N;
    c.V = safe_normalize3(s.V);
    c.R = safe_normalize3(reflect3(-c.V, c.N));
    c.kd = eval_tc(sc, m.kd, m.kd_tex, s.uvw);
    c.ks = eval_tc(sc, m.ks, SPECTEX ? m.ks_tex : -1, s.uvw);
    c.kr = eval_tc(sc, m.kr, SPECTEX ? m.kr_tex : -1, s.uvw);
    c.kt = eval_tc(sc, m.kt, SPECTEX ? m.kt_tex : -1, s.uvw);
    c.alpha = fmaxf(0.001f, eval_tf(sc, m.gloss, SPECTEX ? m.gloss_tex : -1, s.uvw));
    c.ior = m.ior;
    finish_phong_ctx(c, s.front);
    return c;
}
// true when the raw inputs above are the material's constants (no texture on any of them): what nee_kernel can read itself
template <bool SPECTEX = true>
__device__ __forceinline__ bool phong_inputs_constant(const DevMaterial& m) {
    return m.kd_tex < 0 && (!SPECTEX || (m.ks_tex < 0 && m.kr_tex < 0 && m.kt_tex < 0 && m.gloss_tex < 0));
}

// Shading.cpp:280-531
__device__ __forceinline__ bool sample_phong_blinn(const PhongCtx& ctx, bool phong, const SampCtx& s, f3& dir, DirInfo& si) {
    set_void(si);
    const float pd = luma1(ctx.kd);
    const float ps = luma1(ctx.specColor);
    const float pr = luma1(ctx.krf);
    const float pt = luma1(ctx.transColor);
    const float total = pd + ps + pr + pt;
    if (total <= 0.0f) return false;
    const float r = s.rnd(DIM_LOBE) * total;
    const float a = ctx.alpha;
    const float u1 = s.rnd(DIM_U1);
    const float u2 = s.rnd(DIM_U2);

    if (r < pd) {
        f3 U, Vb;
        build_onb(ctx.N, U, Vb);
        const float rxy = sqrtf(u1);
        const float phi = 2.0f * PI_F * u2;
        const float x = rxy * rtx_cosf(phi);
        const float y = rxy * rtx_sinf(phi);
        const float z = sqrtf(fmaxf(0.0f, 1.0f - u1));
        dir = safe_normalize3(U * x + Vb * y + ctx.N * z);
        const float cosN = fmaxf(0.0f, dot3(ctx.N, dir));
        const float pdf = cosN / PI_F;
        si.mult = ctx.kd * (1.0f / PI_F) * cosN;
        si.prob = (pd / total) * pdf;
        si.dist = 0.0f;
        si.lobe = LOBE_DIFFUSE;
        return si.prob > 0.0f && max3c(si.mult) > 0.0f;
    } else if (r < pd + ps || r < pd + ps + pr) {
        const bool specLobe = (r < pd + ps);
        f3 color = ctx.specColor;
        float pl = ps;
        if (!specLobe) {
            color = clamp_max3(ctx.krf, 1.0f);
            pl = pr;
        }
        const float phi = 2.0f * PI_F * u1;
        const float cosT = rtx_powf(u2, 1.0f / (a + 1.0f));
        const float sinT = sqrtf(fmaxf(0.0f, 1.0f - cosT * cosT));
        if (phong) {
            f3 U, Vb;
            build_onb(ctx.R, U, Vb);
            dir = safe_normalize3(U * (rtx_cosf(phi) * sinT) + Vb * (rtx_sinf(phi) * sinT) + ctx.R * cosT);
            const float cosR = fmaxf(0.0f, dot3(ctx.R, dir));
            const float cosN = fmaxf(0.0f, dot3(ctx.N, dir));
            const float pdfDir = ((a + 1.0f) / (2.0f * PI_F)) * rtx_powf(cosR, a);
            if (specLobe) si.mult = color * ((a + 2.0f) / (2.0f * PI_F)) * rtx_powf(cosR, a) * cosN;
            else {
                const float brdf = ((a + 2.0f) / (2.0f * PI_F)) * rtx_powf(cosR, a);
                si.mult = color * brdf;
            }
            si.prob = (pl / total) * pdfDir;
        } else {
            f3 U, Vb;
            build_onb(ctx.N, U, Vb);
            const f3 h = safe_normalize3(U * (rtx_cosf(phi) * sinT) + Vb * (rtx_sinf(phi) * sinT) + ctx.N * cosT);
            float VoH;
            if (specLobe) {
                dir = safe_normalize3(reflect3(-ctx.V, h));
                VoH = fmaxf(0.0f, dot3(ctx.V, h));
            } else {
                VoH = dot3(ctx.V, h);
                dir = safe_normalize3(2.0f * VoH * h - ctx.V);
            }
            const float NoL = fmaxf(0.0f, dot3(ctx.N, dir));
            const float NoH = fmaxf(0.0f, dot3(ctx.N, h));
            if (NoL <= 0.0f || NoH <= 0.0f || VoH <= 1e-6f) return false;
            const float pdfH = ((a + 1.0f) / (2.0f * PI_F)) * rtx_powf(NoH, a);
            const float pdfDir = pdfH / (4.0f * fmaxf(1e-6f, VoH));
            const float invVoH = 1.0f / fmaxf(1e-6f, VoH);
            si.mult = color * ((a + 8.0f) / (8.0f * PI_F)) * rtx_powf(NoH, a) * invVoH;
            si.prob = (pl / total) * pdfDir;
        }
        si.dist = 0.0f;
        si.lobe = LOBE_SPECULAR;
        return si.prob > 0.0f && max3c(si.mult) > 0.0f;
    } else {
        const float eta = s.front ? 1.0f / ctx.ior : ctx.ior;
        const f3 I = -ctx.V;
        const float phi = 2.0f * PI_F * u1;
        const float cosTheta = rtx_powf(u2, 1.0f / (a + 1.0f));
        const float sinTheta = sqrtf(fmaxf(0.0f, 1.0f - cosTheta * cosTheta));
        if (phong) {
            const float cosI = -dot3(ctx.N, I);
            const float k = 1.0f - eta * eta * (1.0f - cosI * cosI);
            if (k < 0.0f) {
                dir = safe_normalize3(reflect3(-ctx.V, ctx.N));
                si.mult = ctx.transColor;
                si.prob = (pt / total);
                si.dist = -1.0f;
                si.lobe = LOBE_SPECULAR;
                return si.prob > 0.0f && max3c(si.mult) > 0.0f;
            }
            const float cosTIdeal = sqrtf(k);
            const f3 T = safe_normalize3(eta * I + (eta * cosI - cosTIdeal) * ctx.N);
            f3 U, Vb;
            build_onb(T, U, Vb);
            dir = safe_normalize3(U * (rtx_cosf(phi) * sinTheta) + Vb * (rtx_sinf(phi) * sinTheta) + T * cosTheta);
            const float cosT = fmaxf(0.0f, dot3(T, dir));
            const float absCosN = fabsf(dot3(ctx.N, dir));
            if (cosT <= 0.0f || absCosN <= 0.0f) return false;
            const float pdfDir = ((a + 1.0f) / (2.0f * PI_F)) * rtx_powf(cosT, a);
            const float etaScale = transmission_scale(eta);
            const f3 fTrans = ctx.transColor * ((a + 2.0f) / (2.0f * PI_F)) * rtx_powf(cosT, a);
            si.mult = fTrans * etaScale;
            si.prob = (pt / total) * pdfDir;
        } else {
            f3 U, Vb;
            build_onb(ctx.N, U, Vb);
            f3 h = safe_normalize3(U * (rtx_cosf(phi) * sinTheta) + Vb * (rtx_sinf(phi) * sinTheta) + ctx.N * cosTheta);
            if (dot3(ctx.V, h) < 0.0f) h = -h;
            const float cosI = -dot3(h, I);
            const float k = 1.0f - eta * eta * (1.0f - cosI * cosI);
            if (k < 0.0f) {
                dir = safe_normalize3(reflect3(-ctx.V, h));
                si.mult = ctx.transColor;
                si.prob = (pt / total);
                si.dist = -1.0f;
                si.lobe = LOBE_SPECULAR;
                return si.prob > 0.0f && max3c(si.mult) > 0.0f;
            }
            const float cosT = sqrtf(k);
            dir = safe_normalize3(eta * I + (eta * cosI - cosT) * h);
            const float absCosN = fabsf(dot3(ctx.N, dir));
            const float VoH = dot3(ctx.V, h);
            const float DoH = dot3(dir, h);
            if (absCosN <= 0.0f || VoH <= 1e-6f || VoH * DoH >= 0.0f) return false;
            const float denom = eta * VoH + DoH;
            if (fabsf(denom) <= 1e-6f) return false;
            const float pdfH = ((a + 1.0f) / (2.0f * PI_F)) * rtx_powf(fmaxf(0.0f, dot3(ctx.N, h)), a);
            const float dwh_dwi = fabsf((eta * eta * DoH) / (denom * denom));
            const float pdfDir = pdfH * dwh_dwi;
            if (pdfDir <= 0.0f) return false;
            const float etaScale = transmission_scale(eta);
            const float NoH = fmaxf(0.0f, dot3(ctx.N, h));
            const f3 fTrans = ctx.transColor * ((a + 2.0f) / (2.0f * PI_F)) * rtx_powf(NoH, a);
            const float factor = fabsf(DoH * VoH) / fmaxf(1e-6f, fabsf(dot3(ctx.N, ctx.V)) * denom * denom);
            si.mult = fTrans * etaScale * factor;
            si.prob = (pt / total) * pdfDir;
        }
        si.dist = 0.0f;
        si.lobe = LOBE_TRANSMISSION;
        return si.prob > 0.0f && max3c(si.mult) > 0.0f;
    }
}

// Shading.cpp:533-713
__device__ __forceinline__ void evaluate_phong_blinn(const PhongCtx& ctx, bool phong, const SampCtx& s, f3 dirIn, DirInfo& si) {
    set_void(si);
    const f3 dir = safe_normalize3(dirIn);
    const float cosN = dot3(ctx.N, dir);
    f3 mult = mk3(0.0f);
    float pdf = 0.0f;
    const float a = ctx.alpha;
    const float pd = luma1(ctx.kd);
    const float ps = luma1(ctx.specColor);
    const float pt = luma1(ctx.transColor);
    int lobe = LOBE_NONE;

    if (cosN > 0.0f) {
        const float pr = luma1(ctx.krf);
        const float total = pd + ps + pr + pt;
        if (total <= 0.0f) return;
        const float pdfDiff = pd > 0.0f ? (cosN / PI_F) * (pd / total) : 0.0f;
        float pdfRefl = 0.0f;
        f3 fRefl = mk3(0.0f);
        float pdfSpec = 0.0f;
        bool hasSpec = false;
        f3 fSpec = mk3(0.0f);
        const f3 fDiff = (pd > 0.0f) ? ctx.kd * (1.0f / PI_F) : mk3(0.0f);
        if (phong) {
            const float cosR = fmaxf(0.0f, dot3(ctx.R, dir));
            if (pr > 0.0f && cosR > 0.0f) {
                const f3 reflColor = clamp_max3(ctx.krf, 1.0f);
                pdfRefl = ((a + 1.0f) / (2.0f * PI_F)) * rtx_powf(cosR, a);
                pdfRefl *= pr / total;
                fRefl = reflColor * ((a + 2.0f) / (2.0f * PI_F)) * rtx_powf(cosR, a);
            }
            if (ps > 0.0f && cosR > 0.0f) {
                pdfSpec = ((a + 1.0f) / (2.0f * PI_F)) * rtx_powf(cosR, a);
                pdfSpec *= ps / total;
                hasSpec = true;
                fSpec = ctx.specColor * ((a + 2.0f) / (2.0f * PI_F)) * rtx_powf(cosR, a);
            }
            mult = fRefl + (fDiff + fSpec) * cosN;
        } else {
            const f3 H = safe_normalize3(ctx.V + dir);
            const float NoH = fmaxf(0.0f, dot3(ctx.N, H));
            const float VoH = fmaxf(0.0f, dot3(ctx.V, H));
            if (NoH > 0.0f && VoH > 1e-6f) {
                const float powNoH = rtx_powf(NoH, a);
                if (pr > 0.0f) {
                    const f3 reflColor = clamp_max3(ctx.krf, 1.0f);
                    const float pdfH = ((a + 1.0f) / (2.0f * PI_F)) * powNoH;
                    const float pdfDir = pdfH / (4.0f * fmaxf(1e-6f, VoH));
                    pdfRefl = pdfDir * (pr / total);
                    const float invVoH = 1.0f / fmaxf(1e-6f, VoH);
                    fRefl = reflColor * ((a + 8.0f) / (8.0f * PI_F)) * powNoH * invVoH;
                }
                if (ps > 0.0f) {
                    float pdfDir = ((a + 1.0f) / (2.0f * PI_F)) * powNoH;
                    pdfDir /= (4.0f * fmaxf(1e-6f, VoH));
                    pdfSpec = pdfDir * (ps / total);
                    hasSpec = true;
                    const float invVoH = 1.0f / fmaxf(1e-6f, VoH);
                    fSpec = ctx.specColor * ((a + 8.0f) / (8.0f * PI_F)) * powNoH * invVoH;
                }
            }
            mult = (fRefl + fDiff + fSpec) * cosN;
        }
        pdf = pdfDiff + pdfSpec + pdfRefl;
        const bool hasRefl = (pdfRefl > 0.0f);
        if ((hasSpec || hasRefl) && pd > 0.0f) lobe = LOBE_ALL;
        else if (hasSpec || hasRefl) lobe = LOBE_SPECULAR;
        else lobe = LOBE_DIFFUSE;
    } else {
        const float pr = ctx.krPos ? luma1(ctx.krf) : 0.0f;
        const float total = pd + ps + pr + pt;
        if (total <= 0.0f) return;
        if (pt > 0.0f) {
            const float eta = s.front ? 1.0f / ctx.ior : ctx.ior;
            const float absCosWo = fabsf(dot3(ctx.N, ctx.V));
            const float absCosWi = fabsf(dot3(ctx.N, dir));
            if (absCosWi <= 0.0f || absCosWo <= 0.0f) return;
            if (phong) {
                const f3 I = -ctx.V;
                const float cosI = -dot3(ctx.N, I);
                const float k = 1.0f - eta * eta * (1.0f - cosI * cosI);
                if (k < 0.0f) return;
                const float cosTIdeal = sqrtf(k);
                const f3 T = safe_normalize3(eta * I + (eta * cosI - cosTIdeal) * ctx.N);
                const float cosT = fmaxf(0.0f, dot3(T, dir));
                if (cosT <= 0.0f) return;
                const float pdfDir = ((a + 1.0f) / (2.0f * PI_F)) * rtx_powf(cosT, a);
                const float etaScale = transmission_scale(eta);
                const f3 fTrans = ctx.transColor * ((a + 2.0f) / (2.0f * PI_F)) * rtx_powf(cosT, a);
                mult = fTrans * etaScale;
                pdf = (pt / total) * pdfDir;
                lobe = LOBE_TRANSMISSION;
            } else {
                f3 H = safe_normalize3(ctx.V + dir * eta);
                if (dot3(ctx.N, H) < 0.0f) H = -H;
                const float NoH = fmaxf(0.0f, dot3(ctx.N, H));
                if (NoH <= 0.0f) return;
                const float VoH = dot3(ctx.V, H);
                const float WiH = dot3(dir, H);
                if (VoH * WiH >= 0.0f) return;
                const float denom = eta * VoH + WiH;
                if (fabsf(denom) <= 1e-6f) return;
                const float pdfH = ((a + 1.0f) / (2.0f * PI_F)) * rtx_powf(NoH, a);
                const float dwh_dwi = fabsf((eta * eta * WiH) / (denom * denom));
                const float pdfDir = pdfH * dwh_dwi;
                const float etaScale = transmission_scale(eta);
                const f3 fTrans = ctx.transColor * ((a + 2.0f) / (2.0f * PI_F)) * rtx_powf(NoH, a);
                const float factor = fabsf(WiH * VoH) / fmaxf(1e-6f, absCosWo * denom * denom);
                mult = fTrans * etaScale * factor;
                pdf = (pt / total) * pdfDir;
                lobe = LOBE_TRANSMISSION;
            }
        }
    }
    if (pdf <= 0.0f || max3c(mult) <= 0.0f) return;
    si.mult = mult;
    si.prob = pdf;
    si.dist = 0.0f;
    si.lobe = lobe;
}

// ---- MtlMicrofacet (Shading.cpp:741-936; materials of FBX imports) -------------------------------------------
struct MicroCtx {
    f3 N, V, F0, diffuse, trans;
    float alpha, ior, pd, ps, pt, total;
};
__device__ __forceinline__ float ggx_d(float alpha2, float NoH) {   // :741-745
    const float denom = NoH * NoH * (alpha2 - 1.0f) + 1.0f;
    return alpha2 / (PI_F * denom * denom);
}
__device__ __forceinline__ float ggx_g1(float alpha2, float NoX) {   // :747-751
    const float denom = NoX + sqrtf(alpha2 + (1.0f - alpha2) * NoX * NoX);
    return (2.0f * NoX) / denom;
}
__device__ __forceinline__ f3 fresnel_schlick(f3 F0, float VoH) {   // :758-762
    const float k = rtx_powf(1.0f - VoH, 5.0f);
    return F0 + (mk3(1.0f) - F0) * k;
}
// common prologue of GenerateSample / GetSampleInfo (:768-785, :861-879), built once per hit; material slots as
// described at RTX_MTL_MICROFACET in rtx.h
__device__ __forceinline__ MicroCtx build_micro_ctx(const DevScene& sc, const DevMaterial& m, const SampCtx& s) {
    MicroCtx c;
    c.N = normalized3(s.N);
    if (!s.front) c.N = -c.N;
    c.V = safe_normalize3(s.V);
    const f3 base = eval_tc(sc, m.kd, m.kd_tex, s.uvw);
    float rough = m.ks[0], metal = m.kr[0];
    if (m.ks_tex >= 0) rough *= tex_eval(sc, m.ks_tex, s.uvw).x;
    if (m.kr_tex >= 0) metal *= tex_eval(sc, m.kr_tex, s.uvw).x;
    rough = clamp01f(rough);
    metal = clamp01f(metal);
    c.trans = eval_tc(sc, m.kt, m.kt_tex, s.uvw);
    c.alpha = fmaxf(0.002f, rough * rough);
    c.F0 = mk3(0.04f) * (1.0f - metal) + base * metal;
    c.diffuse = base * (1.0f - metal);
    c.ior = m.ior;
    c.pd = luma1(c.diffuse);
    c.ps = luma1(c.F0);
    c.pt = luma1(c.trans);
    c.total = c.pd + c.ps + c.pt;
    return c;
}
__device__ __forceinline__ bool sample_microfacet(const MicroCtx& c, const SampCtx& s, f3& dir, DirInfo& si) {   // :765-856
    set_void(si);
    if (c.total <= 0.0f) return false;
    const f3 N = c.N, V = c.V;
    const float r = s.rnd(DIM_LOBE) * c.total;
    const float u1 = s.rnd(DIM_U1), u2 = s.rnd(DIM_U2);
    if (r < c.pd) {
        f3 U, Vb;
        build_onb(N, U, Vb);
        const float rxy = sqrtf(u1);
        const float phi = 2.0f * PI_F * u2;
        const float x = rxy * rtx_cosf(phi);
        const float y = rxy * rtx_sinf(phi);
        const float z = sqrtf(fmaxf(0.0f, 1.0f - u1));
        dir = safe_normalize3(U * x + Vb * y + N * z);
        const float cosN = fmaxf(0.0f, dot3(N, dir));
        const float pdf = cosN / PI_F;
        si.mult = c.diffuse * (1.0f / PI_F) * cosN;
        si.prob = (c.pd / c.total) * pdf;
        si.dist = 0.0f;
        si.lobe = LOBE_DIFFUSE;
        return si.prob > 0.0f && max3c(si.mult) > 0.0f;
    } else if (r < c.pd + c.ps) {
        f3 U, Vb;
        build_onb(N, U, Vb);
        const float phi = 2.0f * PI_F * u1;
        const float cosT = sqrtf((1.0f - u2) / (1.0f + (c.alpha * c.alpha - 1.0f) * u2));
        const float sinT = sqrtf(fmaxf(0.0f, 1.0f - cosT * cosT));
        const f3 h = safe_normalize3(U * (rtx_cosf(phi) * sinT) + Vb * (rtx_sinf(phi) * sinT) + N * cosT);
        dir = reflect3(-V, h);
        if (dot3(N, dir) <= 0.0f) return false;
        const float NoV = fmaxf(0.0f, dot3(N, V));
        const float NoL = fmaxf(0.0f, dot3(N, dir));
        const float NoH = fmaxf(0.0f, dot3(N, h));
        const float VoH = fmaxf(0.0f, dot3(V, h));
        const float alpha2 = c.alpha * c.alpha;
        const float D = ggx_d(alpha2, NoH);
        const float G = ggx_g1(alpha2, NoL) * ggx_g1(alpha2, NoV);
        const f3 F = fresnel_schlick(c.F0, VoH);
        si.mult = (F * (D * G)) / fmaxf(1e-6f, 4.0f * NoV);
        const float pdfDir = (D * NoH) / fmaxf(1e-6f, 4.0f * VoH);
        si.prob = (c.ps / c.total) * pdfDir;
        si.dist = 0.0f;
        si.lobe = LOBE_SPECULAR;
        return si.prob > 0.0f && max3c(si.mult) > 0.0f;
    } else {
        const float eta = s.front ? 1.0f / c.ior : c.ior;
        const f3 I = -V;
        const float cosI = clamp01f(-dot3(N, I));
        const float k = 1.0f - eta * eta * (1.0f - cosI * cosI);
        if (k < 0.0f) return false;
        const f3 T = eta * I + (eta * cosI - sqrtf(k)) * N;
        dir = safe_normalize3(T);
        si.mult = c.trans * transmission_scale(eta);
        si.prob = c.pt / c.total;
        si.dist = -1.0f;   // delta transmission
        si.lobe = LOBE_TRANSMISSION;
        return si.prob > 0.0f && max3c(si.mult) > 0.0f;
    }
}
__device__ __forceinline__ void evaluate_microfacet(const MicroCtx& c, const SampCtx& s, f3 dirIn, DirInfo& si) {   // :858-934
    set_void(si);
    if (c.total <= 0.0f) return;
    const f3 N = c.N, V = c.V;
    const f3 L = safe_normalize3(dirIn);
    const float NoL = dot3(N, L);
    f3 mult = mk3(0.0f);
    float pdf = 0.0f;
    int lobe = LOBE_NONE;
    if (NoL > 0.0f) {
        const f3 fDiff = c.diffuse * (1.0f / PI_F) * NoL;
        const float pdfDiff = (c.pd > 0.0f) ? (NoL / PI_F) * (c.pd / c.total) : 0.0f;
        const f3 H = safe_normalize3(L + V);
        const float NoV = fmaxf(0.0f, dot3(N, V));
        const float NoH = fmaxf(0.0f, dot3(N, H));
        const float VoH = fmaxf(0.0f, dot3(V, H));
        float pdfSpec = 0.0f;
        f3 fSpec = mk3(0.0f);
        if (c.ps > 0.0f && VoH > 0.0f) {
            const float alpha2 = c.alpha * c.alpha;
            const float D = ggx_d(alpha2, NoH);
            const float G = ggx_g1(alpha2, NoL) * ggx_g1(alpha2, NoV);
            const f3 F = fresnel_schlick(c.F0, VoH);
            fSpec = (F * (D * G)) / fmaxf(1e-6f, 4.0f * NoV);
            pdfSpec = (D * NoH) / fmaxf(1e-6f, 4.0f * VoH);
            pdfSpec *= (c.ps / c.total);
        }
        mult = fDiff + fSpec;
        pdf = pdfDiff + pdfSpec;
        lobe = (c.ps > 0.0f && c.pd > 0.0f) ? LOBE_ALL : (c.ps > 0.0f ? LOBE_SPECULAR : LOBE_DIFFUSE);
    } else if (c.pt > 0.0f) {
        const float eta = s.front ? 1.0f / c.ior : c.ior;
        const f3 I = -V;
        const float cosI = -dot3(N, I);
        const float k = 1.0f - eta * eta * (1.0f - cosI * cosI);
        if (k < 0.0f) return;
        const float cosT = sqrtf(k);
        const f3 T = safe_normalize3(eta * I + (eta * cosI - cosT) * N);
        if (dot3(T, L) < 1.0f - 1e-4f) return;   // only the delta refraction direction counts
        mult = c.trans * transmission_scale(eta);
        pdf = c.pt / c.total;
        lobe = LOBE_TRANSMISSION;
    }
    if (pdf <= 0.0f || max3c(mult) <= 0.0f) return;
    si.mult = mult;
    si.prob = pdf;
    si.dist = 0.0f;
    si.lobe = lobe;
}
// IsPhotonSurface stand-in: some diffuse reflectance
__device__ __forceinline__ bool mtl_is_photon_surface(const DevMaterial& m) {
    const float k = m.type == RTX_MTL_MICROFACET ? 1.0f - clamp01f(m.kr[0]) : 1.0f;
    return (m.kd[0] + m.kd[1] + m.kd[2]) * k > 0.0f;
}

// ---- PointLight -------------------------------------------------------------------------------------------
// Shading.cpp:118-165, in two parts so that the NEE kernel (kernels_nee.hip) can keep the direction of a sample and form its
// weights later: light_sample_dir = the sampled point (direction, distance; false: no sample), light_sample_terms = mult and
// prob of a sample at that distance. light_generate_sample is the two together, in the reference's order of operations.
__device__ __forceinline__ bool light_sample_dir(const DevLight& L, f3 P, float u1, float u2, f3& dir, float& distOut) {
    const f3 position = ld3(L.pos);
    const f3 toC = position - P;
    const float dc = len3(toC);
    if (dc < 1e-6f) return false;
    if (L.size <= 0.0f) {
        dir = toC / dc;
        distOut = dc;
        return true;
    }
    const float z = 1.0f - 2.0f * u1;
    const float rxy = sqrtf(fmaxf(0.0f, 1.0f - z * z));
    const float phi = 2.0f * PI_F * u2;
    const f3 onS = mk3(rxy * rtx_cosf(phi), rxy * rtx_sinf(phi), z);
    const f3 Lpos = position + L.size * onS;
    const f3 wi = Lpos - P;
    const float dist = len3(wi);
    if (dist < 1e-6f) return false;
    dir = wi / dist;
    const float cosLight = fmaxf(0.0f, dot3(-onS, dir));
    if (cosLight <= 0.0f) return false;
    distOut = dist;
    return true;
}
__device__ __forceinline__ void light_sample_terms(const DevLight& L, float dist, DirInfo& si) {
    if (L.size <= 0.0f) {
        const f3 intensity = ld3(L.intensity);
        const float att_factor = (L.attenuation == 0.0f) ? 1.0f : L.attenuation * L.attenuation;
        si.mult = intensity / (att_factor * dist * dist);
        si.prob = 1.0f;
    } else {
        si.mult = ld3(L.radAtt);   // radiance / att_factor with radiance = intensity / (pi size^2): DevLight, layout.h
        si.prob = L.pdfArea;       // 2 / (4 pi size^2)
    }
    si.dist = dist;
    si.lobe = LOBE_ALL;
}
__device__ __forceinline__ bool light_generate_sample(const DevLight& L, f3 P, float u1, float u2, f3& dir, DirInfo& si) {
    set_void(si);
    float dist;
    if (!light_sample_dir(L, P, u1, u2, dir, dist)) return false;
    light_sample_terms(L, dist, si);
    return si.prob > 0.0f;
}

// Shading.cpp:167-209
__device__ __forceinline__ void light_get_sample_info(const DevLight& L, f3 P, f3 dirIn, DirInfo& si) {
    set_void(si);
    const f3 position = ld3(L.pos), intensity = ld3(L.intensity);
    const f3 d = safe_normalize3(dirIn);
    if (L.size <= 0.0f) {
        const f3 toL = position - P;
        const float dist = len3(toL);
        if (dist < 1e-6f) return;
        if (dot3(safe_normalize3(toL), d) < 1.0f - 1e-4f) return;
        const float att_factor = (L.attenuation == 0.0f) ? 1.0f : L.attenuation * L.attenuation;
        si.mult = intensity / (att_factor * dist * dist);
        si.prob = 1.0f;
        si.dist = dist;
        si.lobe = LOBE_ALL;
        return;
    }
    const f3 oc = P - position;
    const float b = dot3(oc, d);
    const float c = len2_3(oc) - L.size * L.size;
    const float disc = b * b - c;
    if (disc < 0.0f) return;
    const float t = -b - sqrtf(fmaxf(0.0f, disc));
    if (t <= 1e-6f) return;
    const f3 hit = P + d * t;
    const f3 nL = safe_normalize3(hit - position);
    const float cosLight = fmaxf(0.0f, dot3(-nL, d));
    if (cosLight <= 0.0f) return;
    si.mult = ld3(L.radAtt);
    si.prob = L.pdfArea;
    si.dist = t;
    si.lobe = LOBE_ALL;
}

// PointLight::IntersectRay normal for the MIS term of NEE (RayCaster.cpp:482-486): returns cosLight or 1 on miss
__device__ __forceinline__ float light_cos_for_nee(const DevLight& L, f3 P, f3 d) {
    if (L.size <= 0.0f) return 1.0f;
    const f3 position = ld3(L.pos);
    const f3 oc = P - position;
    const float b = dot3(oc, d);
    const float c = len2_3(oc) - L.size * L.size;
    const float disc = b * b - c;
    if (disc < 0.0f) return 1.0f;
    float t = -b - sqrtf(fmaxf(0.0f, disc));
    if (t < 0.0f) t = -b + sqrtf(fmaxf(0.0f, disc));
    if (t < 0.0f) return 1.0f;
    const f3 hp = P + d * t;
    const f3 N = safe_normalize3(hp - position);
    // HIT_FRONT passed by the caller is ignored by PointLight::IntersectRay (Shading.cpp:65-86)
    return fmaxf(0.0f, dot3(-N, d));
}

// Shading.cpp:88-116
__device__ __forceinline__ void light_random_photon(const DevLight& L, const float u[4], f3& rp, f3& rd, f3& c) {
    const f3 position = ld3(L.pos);
    f3 pos = position;
    int k = 0;
    if (L.size > 0.0f) {
        const float u1 = u[k++];
        const float u2 = u[k++];
        const float z = 1.0f - 2.0f * u1;
        const float rxy = sqrtf(fmaxf(0.0f, 1.0f - z * z));
        const float phi = 2.0f * PI_F * u2;
        pos = position + L.size * mk3(rxy * rtx_cosf(phi), rxy * rtx_sinf(phi), z);
    }
    const float u1 = u[k++];
    const float u2 = u[k++];
    const float z = 1.0f - 2.0f * u1;
    const float rxy = sqrtf(fmaxf(0.0f, 1.0f - z * z));
    const float phi = 2.0f * PI_F * u2;
    f3 dir = mk3(rxy * rtx_cosf(phi), rxy * rtx_sinf(phi), z);
    if (L.size > 0.0f) {
        const f3 n = safe_normalize3(pos - position);
        if (dot3(dir, n) < 0.0f) dir = -dir;
    }
    rp = pos;
    rd = dir;
    c = ld3(L.intensity) * 4.0f * PI_F;
}
