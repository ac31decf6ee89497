// Legacy shading mode on gfx950: the reference's SHADING_USE_LEGACY=1 build (RayTracer/ShadingConfig.h:7-13), i.e.
// TraceLegacy (RayTracer/RayCaster.cpp:579-631) -> MtlBlinn/MtlPhong/MtlMicrofacet::Shade
// (RayTracer/Shading_Legacy.cpp:366-694) with PointLight::Illuminate (:115-210) and the shadow / secondary rays of
// ShadeInfoAdapter (RayTracer/ShadeInfoAdapter.h:49-231).
//
// The reference evaluates a recursion tree per camera sample: every Shade() may call TraceSecondaryRay up to three
// times (diffuse bounce, glossy reflection, glossy refraction), at most 5 levels deep (CanBounce, :44), and one
// TraceShadowRay per light. Here a lane evaluates the tree without recursion but in the reference's own order and
// with its own arithmetic: an open Shade() is a FRAME (LgFrame: the running `local`, the factors its children's colours
// will be multiplied by, where its result goes), the rays it still has to trace are CALLS on a small private stack
// (LgCall: a secondary ray, or the shadow query of one light with that light's terms of `local`), pushed in reverse so
// that they are traced as the recursion would - lights in order, diffuse bounce, reflection, refraction - and a frame is
// closed (its result formed exactly as Shade() returns it, and handed to the frame below) when the stack is back at the
// height it was opened at. legacy_kernel runs the lanes of a wave as state machines that share their traversal and
// shading steps (see there). Per-sample radiance is the oracle's bit for bit.
//
// Random numbers: Shade() number `node` of the tree (root 0; children 3*node+1 diffuse, +2 reflection, +3 refraction)
// draws from the counter generator keyed (pixel, sample) x node x draw slot, identically in oracle/oracle_legacy.cpp.
#define RTX_TRAV_SLAB_TWO_OPS 1   // traverse.h trav_inner_step: see there
#define RTX_RESOLVE_TWO_ARRAYS 1 // surface.h resolve_triangle: edges and words from the TriRec, not from its copy in TriAttr (the register
                                 // allocation of this kernel with the one-line form puts a spill store before a lane re-enable: tools/check_spill_exec.py)
#include "shade_body.h"

#define LG_MAX_BOUNCES 5        // ShadeInfoAdapter::kMaxSpecularBounces (:245)
#define LG_FRAMES (LG_MAX_BOUNCES + 1)   // Shade() calls open at once: bounce 0 .. 5
#define LG_STACK 20             // 3 calls at depth 0, then 2 left + 3 new per level down to depth 4: 11; + one queued
                                // shadow query per light (<= 8) of the Shade() just opened
#define LG_SHADOW_CALL 0x80000000u   // LgCall::kind of a queued TraceShadowRay query (bit 0: the light has a second term)
#define LG_DIM_REFL 32          // draw slots of SampleGlossy for the reflection / refraction ray
#define LG_DIM_REFR 34
#ifndef LG_BATCH
#define LG_BATCH 48u            // finished rays of a wave before it stops traversing to complete (shade) them: completion
                                // is the long divergent part, so it pays to run it for many lanes at once (4 spp frame:
                                // 527 / 330 / 261 / 203 ms at 1 / 8 / 16 / 32 with six waves per SIMD)
#endif
#ifndef LG_LEAF_WEIGHT
#define LG_LEAF_WEIGHT 1u
#endif

// Calls and frames live in global memory, one contiguous block per lane of the persistent grid (launch_legacy's
// `state`): a call is one 64-byte line, a frame's everyday fields are one line. In compiler scratch a private array is
// interleaved dword by dword over the lanes of a wave, and once the lanes' stack heights differ every dword of a call
// becomes a 64-byte request of its own (first frame version: 2.6x the L2 writes of the flattened kernel).
struct alignas(16) LgCall {     // one pending TraceSecondaryRay call or TraceShadowRay query
    f3 o;                       // biased origin (:99-112 / :57-70)
    float selfDist;             // shadow query: hits of the receiver closer than this are ignored
    f3 d;                       // unit direction
    float limit;                // shadow query: blocked before `limit` (< 0: unbounded segment; -2: TraceShadowRay returns 1 whatever is hit)
    f3 a;                       // secondary ray: Beer-Lambert coefficients over the call's `dist` (refraction rays), else 0
    uint32_t node;              // secondary ray: RNG node of the Shade() the call leads to; shadow query: the receiver's object
    f3 b;                       // shadow query: a, b = the light's terms of `local`, added in this order if nothing blocks
    uint32_t kind;              // secondary ray: bounce of that Shade() | diffuse depth of the caller << 8 | slot << 16
                                // (1 diffuse, 2 reflection, 3 refraction: where the colour goes in the caller's frame)
};

#define LG_F_SLOT 3u            // frame flags: slot of the frame below that receives the result (0: the sample itself)
#define LG_F_MICRO 4u
#define LG_F_SPAWN 8u
#define LG_F_DIEL 16u
#define LG_F_INDIRECT 32u       // the fields below hold something (else they were never written and read as zero / one)
#define LG_F_CAUSTIC 64u
#define LG_F_EMISSION 128u
#define LG_F_FACTOR 256u
#define LG_F_REFLC 512u
#define LG_F_REFRC 1024u
// One open Shade(). The frame is written and read field by field, and only the fields a Shade() needs (a diffuse
// surface: diffMult, diffProb, indirect): 2.4 KB of calls and frames per lane is more than any cache holds for a whole
// grid, and the kernel's time follows that traffic.
struct alignas(16) LgFrame {
    f3 local;                   // `local` so far (microfacet: `result`), base, flags: valid while a deeper frame is open
    uint32_t base;              //   (the innermost frame's copies are LgState::topLocal / topBase / topFlags)
    f3 diffMult;                // diffuse bounce: indirect = colour * diffMult / diffProb (microfacet: colour * diffMult)
    float diffProb;
    f3 indirect;                // global-map term, or the diffuse bounce once its colour is in (:419-457)
    uint32_t flags;
    f3 caustic;
    uint32_t pad0;              // ---- 64 bytes: all a diffuse surface touches ----
    f3 oneMinusRefl;            // :499-507
    uint32_t pad1;
    f3 reflMul;
    uint32_t pad2;
    f3 refrMul;
    uint32_t pad3;
    f3 reflC;                   // colours of the reflection / refraction call
    uint32_t pad4;
    f3 refrC;
    uint32_t pad5;
    f3 emission;
    uint32_t pad6;
    f3 retFactor;               // Beer-Lambert factor on the result (a refraction call's hit, :359)
    uint32_t pad7;
    uint32_t pad8[4];
};
static_assert(sizeof(LgCall) == 64 && sizeof(LgFrame) == 192, "one call = one 64-byte line, one frame = three");
#define LG_LANE_BYTES (LG_STACK * sizeof(LgCall) + LG_FRAMES * sizeof(LgFrame))

struct LgState {
    f3 L;                       // the sample's radiance once the root frame has closed
    uint32_t sp, fp, rays, shadows;
    // the innermost open frame's running sum, base and flags live here, in registers: three of four returning calls are
    // shadow queries, which then touch no memory at all
    f3 topLocal;
    uint32_t topBase;           // height of the call stack when the frame was opened
    uint32_t topFlags;
    LgCall* stack;              // LG_STACK entries
    LgFrame* frames;            // LG_FRAMES entries
};

// PLAIN: Blinn materials only and no photon maps in this call (launch_legacy decides): the Phong and microfacet forms of
// Shade() and the two photon gathers are then not compiled in (the same specialisation as shade_kernel's)
template <class SC, bool PLAIN = false>
struct LgSceneT {
    static constexpr bool kPlain = PLAIN;
    const SC& sc;
    const DevPhotonMap& pmGlobal;
    const DevPhotonMap& pmCaustic;
    TravStack st;
};

__device__ __forceinline__ f3 clamp01_3(f3 c) { return mk3(clamp01f(c.x), clamp01f(c.y), clamp01f(c.z)); }

// Shading_Legacy.cpp:321-329 as a factor on the colour
__device__ __forceinline__ f3 lg_absorption_factor(f3 absorption, float dist) {
    if (dist <= 0.0f) return mk3(1.0f);
    if (absorption.x == 0.0f && absorption.y == 0.0f && absorption.z == 0.0f) return mk3(1.0f);
    return mk3(rtx_expf(-absorption.x * dist), rtx_expf(-absorption.y * dist), rtx_expf(-absorption.z * dist));
}

// PointLight::IntersectRay of the legacy build (Shading_Legacy.cpp:212-235): distance or -1
__device__ __forceinline__ float lg_light_intersect(const DevLight& L, f3 o, f3 d) {
    if (L.size <= 0.0f) return -1.0f;
    const f3 toC = ld3(L.pos) - o;
    const float tca = dot3(toC, d);
    if (tca < 0.0f) return -1.0f;
    const float d2 = len2_3(toC) - tca * tca;
    if (d2 > L.size * L.size) return -1.0f;
    const float thc = sqrtf(L.size * L.size - d2);
    float t0 = tca - thc, t1 = tca + thc;
    if (t0 > t1) {
        const float t = t0;
        t0 = t1;
        t1 = t;
    }
    if (t0 < 0.0f) t0 = t1;
    if (t0 < 0.0f) return -1.0f;
    return t0;
}

// ShadeInfoAdapter::TraceShadowRay (:49-88) as a queued call: one closest hit from the biased origin, traced by the
// shared traversal loop; lg_shadow_visible decides it from that hit - the receiver's own node is ignored close to the
// origin, a finite segment is blocked only before the light. The result is exactly 0 or 1, so the query carries the
// light's terms of `local` (t1, then t2 if hasT2) and they are added to the open frame if nothing blocks.
__device__ __forceinline__ void lg_push_shadow(const SampCtx& sx, int obj, f3 dir, float tMax, f3 t1, f3 t2, bool hasT2, LgState& S) {
    if (S.sp >= LG_STACK) return;   // cannot happen (see LG_STACK)
    const float Llen = len3(dir);
    const bool always = Llen < 1e-12f;   // TraceShadowRay returns 1 without tracing; kept in the queue for the order of the sums
    const bool finiteSegment = tMax < RTX_BIG * 0.5f;
    const float segLenWorld = finiteSegment ? Llen * tMax : RTX_BIG;
    const f3 Ldir = always ? mk3(0.0f, 1.0f, 0.0f) : dir / Llen;
    const f3 N = sx.N;
    const float nDotL = dot3(N, Ldir);
    const float baseBias = 1e-4f;
    const float nBias = baseBias * (1.0f + 2.0f * (1.0f - fabsf(nDotL)));
    const float dBias = baseBias * 0.5f;
    LgCall& c = S.stack[S.sp++];
    c.o = sx.P + ((nDotL >= 0.0f) ? N : -N) * nBias + Ldir * dBias;
    c.d = Ldir;
    c.a = t1;
    c.b = t2;
    c.selfDist = (nBias + dBias) * 6.0f;
    c.limit = always ? -2.0f : (finiteSegment ? segLenWorld - nBias * 1.5f : -1.0f);
    c.node = (uint32_t)obj;
    c.kind = LG_SHADOW_CALL | (hasT2 ? 1u : 0u);
    S.shadows++;
}
template <class SC>
__device__ __forceinline__ bool lg_shadow_visible(const SC& sc, const LgCall& c, const HitT& h) {
    if (c.limit == -2.0f) return true;
    if (h.slot < 0) return true;
    const RayT r{c.o.x, c.o.y, c.o.z, c.d.x, c.d.y, c.d.z, RTX_BIG};
    f3 GN;
    bool front;
    int hobj;
    triangle_gn(sc, r, h.slot, h.t, GN, front, hobj);
    const float4 b = reinterpret_cast<const float4*>(sc.tris)[(size_t)h.slot * 3 + 1];
    f3 hp = c.o + c.d * h.t;
    if (__float_as_uint(b.w) != 0u) {   // Sphere / Plane: the hit point comes back through the node transform
        SurfHit sh;
        resolve_triangle(sc, r, h, sh);
        hp = sh.p;
    }
    const float dist = len3(hp - c.o);
    if (hobj == (int)c.node && dist < c.selfDist) return true;
    if (c.limit < 0.0f) return !(dist > 0.0f);
    return !(dist < c.limit);
}
// PointLight::Illuminate (Shading_Legacy.cpp:115-210) with the shadow query queued. TraceShadowRay returns exactly 0 or
// 1, so the first cone sample is always "unanimous" (:189) and the 2nd..4th samples of the adaptive loop are never taken:
// one shadow ray per light, and Li = I * 1 * atten if it arrives. Returns false when Li is zero without a query.
__device__ __forceinline__ bool lg_illuminate_deferred(const SampCtx& sx, const DevLight& light, int l, f3& Li, f3& shadowDir) {
    const f3 toC = ld3(light.pos) - sx.P;
    const float distC = len3(toC);
    if (distC < 1e-20f) return false;
    const f3 L = toC / distC;
    float atten = 1.0f;
    if (light.attenuation > 0.0f) atten = 1.0f / ((light.attenuation * distC) * (light.attenuation * distC));
    Li = ld3(light.intensity) * 1.0f * atten;
    if (light.size <= 0.0f) {
        shadowDir = toC;
        return true;
    }
    f3 U, V;
    build_onb(L, U, V);
    const float jx = sx.rnd(DIM_LIGHT0 + 2 * l), jy = sx.rnd(DIM_LIGHT0 + 2 * l + 1);
    float h2 = 0.5f + jx;                 // Halton(1, 2)
    h2 -= floorf(h2);
    float h3 = (1.0f / 3.0f) * 1.0f + jy;   // Halton(1, 3)
    h3 -= floorf(h3);
    const float gamma = rtx_asinf(light.size / distC);
    const float cosTheta = 1.0f - h2 * (1.0f - rtx_cosf(gamma));
    const float sinTheta = sqrtf(fmaxf(0.0f, 1.0f - cosTheta * cosTheta));
    const float phi = 2.0f * PI_F * h3;
    const f3 sampleDir = normalized3(U * rtx_cosf(phi) * sinTheta + V * rtx_sinf(phi) * sinTheta + L * cosTheta);
    shadowDir = sampleDir * distC;
    return true;
}

// FresnelExact (Shading_Legacy.cpp:296-318)
__device__ __forceinline__ float lg_fresnel_exact(f3 I, f3 N, float ior) {
    float cosi = fminf(fmaxf(dot3(I, N), -1.0f), 1.0f);
    float etai = 1.0f, etat = ior;
    if (cosi > 0.0f) {
        etai = ior;
        etat = 1.0f;
    }
    cosi = fabsf(cosi);
    const float sint = etai / etat * sqrtf(fmaxf(0.0f, 1.0f - cosi * cosi));
    if (sint >= 1.0f) return 1.0f;
    const float cost = sqrtf(fmaxf(0.0f, 1.0f - sint * sint));
    const float Rs = ((etat * cosi) - (etai * cost)) / ((etat * cosi) + (etai * cost));
    const float Rp = ((etai * cosi) - (etat * cost)) / ((etai * cosi) + (etat * cost));
    return clamp01f(0.5f * (Rs * Rs + Rp * Rp));
}

// The ray of a TraceSecondaryRay call (ShadeInfoAdapter.h:99-112), queued; slot: where its colour goes in the open frame
__device__ __forceinline__ void lg_push(LgState& S, const SampCtx& sx, f3 dir, f3 absorb, uint32_t node, int bounce, int diffuseDepth, uint32_t slot) {
    const float dlen = len3(dir);
    if (dlen < 1e-20f) return;                                 // the call returns black: the slot keeps its zero
    if (S.sp >= LG_STACK) return;                              // cannot happen (see LG_STACK)
    dir = dir / dlen;
    const f3 N = sx.N;
    const float nDotD = dot3(N, dir);
    const float baseBias = 1e-4f;
    const float nBias = baseBias * (1.0f + 2.0f * (1.0f - fabsf(nDotD)));
    const float dBias = baseBias * 0.5f;
    LgCall& c = S.stack[S.sp++];
    c.o = sx.P + ((nDotD >= 0.0f) ? N : -N) * nBias + dir * dBias;
    c.d = dir;
    c.a = absorb;   // b, selfDist, limit: shadow queries only
    c.node = node;
    c.kind = (uint32_t)bounce | ((uint32_t)diffuseDepth << 8) | (slot << 16);
}

// a colour arrives in slot `slot` of the innermost frame (no frame: it is the answer itself)
__device__ __forceinline__ void lg_deliver(LgState& S, uint32_t slot, f3 col) {
    if (S.fp == 0u) {
        S.L = col;
        return;
    }
    LgFrame& F = S.frames[S.fp - 1u];
    if (slot == 1u) {
        F.indirect = (S.topFlags & LG_F_MICRO) ? col * F.diffMult : col * F.diffMult / F.diffProb;   // :689 / :443
        S.topFlags |= LG_F_INDIRECT;
    } else if (slot == 2u) {
        F.reflC = col;
        S.topFlags |= LG_F_REFLC;
    } else {
        F.refrC = col;
        S.topFlags |= LG_F_REFRC;
    }
}
// frames whose calls have all returned are closed, innermost first: Shade()'s result (:446-447, :499-512 / :690-692).
// Fields that were never written enter as the zeros (retFactor: the one) the reference computes with there.
__device__ __forceinline__ void lg_unwind(LgState& S) {
    while (S.fp > 0u && S.sp == S.topBase) {
        const LgFrame& F = S.frames[S.fp - 1u];
        const uint32_t fl = S.topFlags;
        // the frame below becomes the innermost one again: its fields are fetched together with F's
        const uint32_t below = S.fp >= 2u ? S.fp - 2u : 0u;
        const f3 belowLocal = S.frames[below].local;
        const uint32_t belowBase = S.frames[below].base, belowFlags = S.frames[below].flags;
        const f3 zero = mk3(0.0f);
        const f3 ind = (fl & LG_F_INDIRECT) ? F.indirect : zero;
        f3 result;
        if (fl & LG_F_MICRO) {
            result = S.topLocal + ind;
        } else {
            const f3 indirect = ind + ((fl & LG_F_CAUSTIC) ? F.caustic : zero);
            const f3 local = S.topLocal + indirect;
            result = local;
            if (fl & LG_F_SPAWN) {
                const f3 reflC = (fl & LG_F_REFLC) ? F.reflC : zero;
                if (fl & LG_F_DIEL) result = local * F.oneMinusRefl + reflC * F.reflMul + ((fl & LG_F_REFRC) ? F.refrC : zero) * F.refrMul;
                else result = local + reflC * F.reflMul;
            }
        }
        result = result + ((fl & LG_F_EMISSION) ? F.emission : zero);
        const f3 out = result * ((fl & LG_F_FACTOR) ? F.retFactor : mk3(1.0f));
        --S.fp;
        S.topLocal = belowLocal;
        S.topBase = belowBase;
        S.topFlags = belowFlags;
        lg_deliver(S, fl & LG_F_SLOT, out);
    }
}

// direction of SampleGlossy (Shading_Legacy.cpp:332-356), glossySamples = 1
__device__ __forceinline__ f3 lg_glossy_dir(const SampCtx& sx, f3 axis, float alpha, bool isReflection, f3 V) {
    f3 U, Vb;
    build_onb(axis, U, Vb);
    const uint32_t d0 = isReflection ? LG_DIM_REFL : LG_DIM_REFR;
    const float u1 = sx.rnd(d0), u2 = sx.rnd(d0 + 1);
    const float cosTheta = rtx_powf(u1, 1.0f / (alpha + 1.0f));
    const float sinTheta = sqrtf(fmaxf(0.0f, 1.0f - cosTheta * cosTheta));
    const float phi = 2.0f * PI_F * u2;
    const f3 H = normalized3(U * rtx_cosf(phi) * sinTheta + Vb * rtx_sinf(phi) * sinTheta + axis * cosTheta);
    const f3 dir = isReflection ? (2.0f * dot3(V, H) * H - V) : H;
    return normalized3(dir);
}

// legacy GenerateSample: MtlBlinn :697-788, MtlPhong :790-874, MtlMicrofacet :876-897
template <bool PLAIN>
__device__ __forceinline__ bool lg_generate_sample(const DevScene& sc, const DevMaterial& m, const SampCtx& s, f3& dir, DirInfo& si) {
    if (!PLAIN && m.type == RTX_MTL_MICROFACET) {
        const f3 N = normalized3(s.N);
        f3 U, Vb;
        build_onb(N, U, Vb);
        const float u1 = s.rnd(DIM_U1), u2 = s.rnd(DIM_U2);
        const float cosTheta = sqrtf(u1);
        const float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
        const float phi = 2.0f * PI_F * u2;
        dir = U * (sinTheta * rtx_cosf(phi)) + Vb * (sinTheta * rtx_sinf(phi)) + N * cosTheta;
        si.mult = eval_tc(sc, m.kd, m.kd_tex, s.uvw);
        si.prob = cosTheta / PI_F;
        si.lobe = LOBE_DIFFUSE;
        return true;
    }
    const bool phong = !PLAIN && m.type == RTX_MTL_PHONG;
    f3 N = normalized3(s.N);
    if (!s.front) N = -N;
    const f3 V = normalized3(s.V);
    const float u1 = s.rnd(DIM_U1), u2 = s.rnd(DIM_U2);
    const f3 diff = eval_tc(sc, m.kd, m.kd_tex, s.uvw);
    const f3 refl = eval_tc(sc, m.kr, m.kr_tex, s.uvw);
    const f3 refr = eval_tc(sc, m.kt, m.kt_tex, s.uvw);
    const float diffWeight = diff.x + diff.y + diff.z, reflWeight = refl.x + refl.y + refl.z, refrWeight = refr.x + refr.y + refr.z;
    const float total = diffWeight + reflWeight + refrWeight;
    if (total < 1e-6f) return false;
    const float pick = s.rnd(DIM_LOBE) * total;
    const float alpha = eval_tf(sc, m.gloss, m.gloss_tex, s.uvw);
    if (pick < diffWeight) {
        f3 U, Vb;
        build_onb(N, U, Vb);
        const float cosTheta = sqrtf(u1);
        const float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
        const float phi = 2.0f * PI_F * u2;
        dir = U * (sinTheta * rtx_cosf(phi)) + Vb * (sinTheta * rtx_sinf(phi)) + N * cosTheta;
        si.mult = diff;
        si.prob = diffWeight / total;
        si.lobe = LOBE_DIFFUSE;
    } else if (pick < diffWeight + reflWeight) {
        f3 R = 2.0f * dot3(N, V) * N - V;
        R = normalized3(R);
        f3 U, Vb;
        build_onb(R, U, Vb);
        const float cosTheta = rtx_powf(u1, 1.0f / (alpha + 1.0f));
        const float sinTheta = sqrtf(fmaxf(0.0f, 1.0f - cosTheta * cosTheta));
        const float phi = 2.0f * PI_F * u2;
        f3 H = U * (sinTheta * rtx_cosf(phi)) + Vb * (sinTheta * rtx_sinf(phi)) + R * cosTheta;
        H = normalized3(H);
        if (phong) dir = H;
        else dir = normalized3(2.0f * dot3(V, H) * H - V);
        si.mult = refl;
        si.prob = reflWeight / total;
        si.lobe = LOBE_SPECULAR;
    } else {
        const f3 I = -V;
        float n1 = 1.0f, n2 = m.ior;
        if (!s.front) {
            n1 = m.ior;
            n2 = 1.0f;
        }
        const float eta = n1 / n2;
        const float cosi = -dot3(N, I);
        const float k = 1.0f - eta * eta * (1.0f - cosi * cosi);
        if (k < 0.0f) return false;
        dir = eta * I + (eta * cosi - sqrtf(k)) * N;
        dir = normalized3(dir);
        const f3 T = dir;
        if (alpha > 0.0f) {
            f3 U, Vb;
            build_onb(T, U, Vb);
            const float cosTheta = rtx_powf(u1, 1.0f / (alpha + 1.0f));
            const float sinTheta = sqrtf(fmaxf(0.0f, 1.0f - cosTheta * cosTheta));
            const float phi = 2.0f * PI_F * u2;
            dir = U * (sinTheta * rtx_cosf(phi)) + Vb * (sinTheta * rtx_sinf(phi)) + T * cosTheta;
            dir = normalized3(dir);
        }
        si.mult = refr;
        si.prob = refrWeight / total;
        si.lobe = LOBE_TRANSMISSION;
    }
    return true;
}

// Opens the frame of one Shade() (Shading_Legacy.cpp:366-514 Blinn, :517-659 Phong, :662-694 microfacet) and queues its
// calls, last first: refraction, reflection, diffuse bounce, then the lights from the last to the first.
template <class G>
__device__ __forceinline__ void lg_shade(const G& g, const DevMaterial& m, const SampCtx& sx, int obj, uint32_t retSlot, f3 retFactor,
                                         int bounce, int diffuseDepth, LgState& S) {
    const auto& sc = g.sc;
    const bool canBounce = bounce < LG_MAX_BOUNCES;
    const uint32_t node = sx.stream;
    if (S.fp >= (uint32_t)LG_FRAMES) return;   // cannot happen: a frame per bounce level
    if (S.fp > 0u) {   // the frame that stops being the innermost one gets its registers back
        LgFrame& P = S.frames[S.fp - 1u];
        P.local = S.topLocal;
        P.base = S.topBase;
        P.flags = S.topFlags;
    }
    LgFrame& F = S.frames[S.fp++];
    S.topLocal = mk3(0.0f);
    S.topBase = S.sp;
    uint32_t fl = retSlot;
    if (!(retFactor.x == 1.0f && retFactor.y == 1.0f && retFactor.z == 1.0f)) {
        F.retFactor = retFactor;
        fl |= LG_F_FACTOR;
    }
    const f3 emission = eval_tc(sc, m.ke, m.ke_tex, sx.uvw);
    if (!(emission.x == 0.0f && emission.y == 0.0f && emission.z == 0.0f)) {
        F.emission = emission;
        fl |= LG_F_EMISSION;
    }
    if (!G::kPlain && m.type == RTX_MTL_MICROFACET) {
        fl |= LG_F_MICRO;
        const f3 N = normalized3(sx.N);
        const f3 base = eval_tc(sc, m.kd, m.kd_tex, sx.uvw);
        if (max3c(base) > 1e-6f && canBounce) {
            f3 indirDir;
            DirInfo si;
            if (lg_generate_sample<G::kPlain>(sc, m, sx, indirDir, si) && !(si.mult.x == 0.0f && si.mult.y == 0.0f && si.mult.z == 0.0f)) {
                F.diffMult = si.mult;
                lg_push(S, sx, indirDir, mk3(0.0f), node * 3u + 1u, bounce + 1, diffuseDepth, 1u);
            }
        }
        S.topFlags = fl;
        if (bounce == 0) {
            for (int l = sc.n_lights - 1; l >= 0; --l) {
                const f3 Ldir = normalized3(normalized3(ld3(sc.lights[l].pos) - sx.P));   // Illuminate's dir, then Ldir.Normalize() (:672)
                const float NdotL = fmaxf(0.0f, dot3(N, Ldir));
                if (!(NdotL > 0.0f)) continue;   // Illuminate's result would be multiplied by nothing
                f3 Li, sdir;
                if (!lg_illuminate_deferred(sx, sc.lights[l], l, Li, sdir)) continue;
                const f3 t1 = base * Li * NdotL;
                if (t1.x == 0.0f && t1.y == 0.0f && t1.z == 0.0f) continue;
                lg_push_shadow(sx, obj, sdir, 1.0f, t1, mk3(0.0f), false, S);
            }
        }
        return;
    }
    const bool phong = !G::kPlain && m.type == RTX_MTL_PHONG;
    f3 N = normalized3(sx.N);
    if (!sx.front) N = -N;
    const f3 V = normalized3(sx.V);
    const f3 I = -V;
    const float alpha = eval_tf(sc, m.gloss, m.gloss_tex, sx.uvw);
    const f3 kd = eval_tc(sc, m.kd, m.kd_tex, sx.uvw);
    const f3 ks = eval_tc(sc, m.ks, m.ks_tex, sx.uvw);
    const f3 refl = eval_tc(sc, m.kr, m.kr_tex, sx.uvw);
    const f3 refrRaw = eval_tc(sc, m.kt, m.kt_tex, sx.uvw);
    const bool isDielectric = max3c(refrRaw) > 1e-6f;
    const bool spawn = (isDielectric || max3c(refl) > 1e-4f) && canBounce;

    // reflection / refraction (:459-508), queued first = traced last
    if (spawn) {
        fl |= LG_F_SPAWN | (isDielectric ? LG_F_DIEL : 0u);
        const float kr = isDielectric ? lg_fresnel_exact(I, N, m.ior) : 1.0f;
        const f3 reflMul = isDielectric ? mk3(kr) : refl;
        F.reflMul = reflMul;
        if (isDielectric) {
            const f3 refrMul = (1.0f - kr) * refrRaw;
            F.oneMinusRefl = mk3(1.0f) - refl;
            F.refrMul = refrMul;
            float n1 = 1.0f, n2 = m.ior;
            const float cosi = -dot3(N, I);
            if (!sx.front) {
                n1 = m.ior;
                n2 = 1.0f;
            }
            const float eta = n1 / n2;
            const float k = 1.0f - eta * eta * (1.0f - cosi * cosi);
            if (k >= 0.0f && !(refrMul.x == 0.0f && refrMul.y == 0.0f && refrMul.z == 0.0f)) {
                f3 T = eta * I + (eta * cosi - sqrtf(k)) * N;
                T = normalized3(T);
                const f3 dir = lg_glossy_dir(sx, T, alpha, false, V);
                lg_push(S, sx, dir, ld3(m.absorption), node * 3u + 3u, bounce + 1, diffuseDepth, 3u);
            }
        }
        if (!(reflMul.x == 0.0f && reflMul.y == 0.0f && reflMul.z == 0.0f)) {
            const f3 rdir = lg_glossy_dir(sx, N, alpha, true, V);
            lg_push(S, sx, rdir, mk3(0.0f), node * 3u + 2u, bounce + 1, diffuseDepth, 2u);
        }
    }

    // indirect light (:419-457)
    const bool hasDiffuse = max3c(kd) > 1e-6f;
    if (!G::kPlain && hasDiffuse && g.pmCaustic.n > 0) {
        F.caustic = (kd / PI_F) * estimate_irradiance(g.pmCaustic, sx.P, sx.N);
        fl |= LG_F_CAUSTIC;
    }
    if (!G::kPlain && g.pmGlobal.n > 0 && !(diffuseDepth <= 1)) {
        F.indirect = mk3(0.0f) + (kd / PI_F) * estimate_irradiance(g.pmGlobal, sx.P, sx.N);
        fl |= LG_F_INDIRECT;
    } else if (hasDiffuse && canBounce) {
        f3 diffDir;
        DirInfo si;
        if (lg_generate_sample<G::kPlain>(sc, m, sx, diffDir, si) && !(si.mult.x == 0.0f && si.mult.y == 0.0f && si.mult.z == 0.0f)) {
            F.diffMult = si.mult;
            F.diffProb = si.prob;
            lg_push(S, sx, diffDir, mk3(0.0f), node * 3u + 1u, bounce + 1, diffuseDepth, 1u);
        }
    }
    S.topFlags = fl;

    // direct light (USE_PHOTON_MAP_DIRECT = false), the last light first
    for (int l = sc.n_lights - 1; l >= 0; --l) {
        const DevLight& light = sc.lights[l];
        const f3 Ldir = normalized3(normalized3(ld3(light.pos) - sx.P));   // Illuminate's dir, then Ldir.Normalize() (:408)
        const float NdotL = fmaxf(0.0f, dot3(N, Ldir));
        if (!(NdotL > 0.0f)) continue;   // the reference calls Illuminate first and drops its result here (:406-410)
        f3 Li, sdir;
        if (!lg_illuminate_deferred(sx, light, l, Li, sdir)) continue;
        f3 t1 = (kd / PI_F) * Li * NdotL, t2 = mk3(0.0f);
        bool hasT2 = false;
        if (!phong) {
            const f3 H = normalized3(Ldir + V);
            const float NdotH = fmaxf(0.0f, dot3(N, H));
            if (NdotH > 0.0f) {
                t2 = (ks * (alpha + 2.0f) / (8.0f * PI_F)) * Li * rtx_powf(NdotH, alpha);
                hasT2 = true;
            }
        } else {
            const f3 R = normalized3(2.0f * dot3(N, Ldir) * N - Ldir);
            const float RdotV = fmaxf(0.0f, dot3(R, V));
            if (RdotV > 0.0f) t1 = t1 + (ks * (alpha + 1.0f) / (2.0f * PI_F)) * Li * rtx_powf(RdotV, alpha);   // lightContrib, added once
        }
        if (t1.x == 0.0f && t1.y == 0.0f && t1.z == 0.0f && t2.x == 0.0f && t2.y == 0.0f && t2.z == 0.0f) continue;
        lg_push_shadow(sx, obj, sdir, 1.0f, t1, t2, hasT2, S);
    }
}

// A traced call returns: c is the call, h the closest hit of its ray (isRoot: the camera ray's hit, traced by extend_kernel).
// Shadow query: the light's terms enter the open frame. Secondary ray: TraceSecondaryRay from the hit on
// (ShadeInfoAdapter.h:114-230) - a renderable light in front of the geometry, the environment, or the Shade() of the
// surface hit, whose frame is opened here. Closes whatever frames this completes.
struct LgRootOut {
    float primaryDist;
    f3 normal, albedo;
    bool hit;        // a surface was hit (normal valid)
    bool shaded;     // ... and it has a material (albedo valid)
};
template <class G>
__device__ __forceinline__ void lg_complete(const G& g, unsigned long long key, const LgCall& c, const HitT& h, bool isRoot, LgState& S,
                                            LgRootOut* ro, float* distOut) {
    const auto& sc = g.sc;
    if (c.kind & LG_SHADOW_CALL) {
        if (lg_shadow_visible(sc, c, h)) {
            S.topLocal = S.topLocal + c.a;
            if (c.kind & 1u) S.topLocal = S.topLocal + c.b;
        }
        lg_unwind(S);
        return;
    }
    const uint32_t slot = (c.kind >> 16) & 3u;
    const RayT r{c.o.x, c.o.y, c.o.z, c.d.x, c.d.y, c.d.z, RTX_BIG};
    if (!isRoot) {   // nearest renderable light against the closest geometry hit
        float bestLightT = RTX_BIG;
        int visibleLight = -1;
        for (int l = 0; l < sc.n_lights; ++l) {
            const float t = lg_light_intersect(sc.lights[l], c.o, c.d);
            if (t > 0.0f && t < bestLightT) {
                bestLightT = t;
                visibleLight = l;
            }
        }
        const bool hitGeom = h.slot >= 0;
        if (visibleLight >= 0 && (!hitGeom || bestLightT < h.t)) {
            const DevLight& light = sc.lights[visibleLight];
            const f3 rad = clamp01_3(ld3(light.intensity) / (PI_F * light.size * light.size));
            if (distOut) *distOut = bestLightT;
            lg_deliver(S, slot, rad * lg_absorption_factor(c.a, bestLightT));
            lg_unwind(S);
            return;
        }
        if (!hitGeom) {
            const f3 nd = normalized3(c.d);
            const f3 env = sc.env_tex >= 0 ? ld3(sc.env) * tex_eval_env(sc, sc.env_tex, nd) : ld3(sc.env);
            lg_deliver(S, slot, env * lg_absorption_factor(c.a, RTX_BIG));
            lg_unwind(S);
            return;
        }
    }
    SurfHit sh;
    resolve_triangle(sc, r, h, sh);
    const float hitDist = len3(sh.p - c.o);
    const float dist = (isRoot || !sh.front) ? 0.0f : hitDist;   // the root carries no absorption
    if (distOut) *distOut = dist;
    const int mi = sc.objs[sh.obj].material;
    if (ro) {
        ro->primaryDist = hitDist;
        ro->normal = safe_normalize3(sh.GN);
        ro->hit = true;
        ro->shaded = mi >= 0;
    }
    if (mi < 0) {
        lg_deliver(S, slot, mk3(0.0f));
        lg_unwind(S);
        return;
    }
    const DevMaterial& m = sc.mtls[mi];
    apply_bump(sc, m, sh);
    SampCtx sx;
    sx.P = sh.p;
    sx.V = -normalized3(c.d);
    sx.N = sh.N;
    sx.GN = sh.GN;
    sx.uvw = sh.uvw;
    sx.front = sh.front;
    sx.key = key;
    sx.stream = c.node;
    if (ro) ro->albedo = (!G::kPlain && m.type == RTX_MTL_MICROFACET) ? mk3(0.5f) : eval_tc(sc, m.kd, m.kd_tex, sx.uvw);   // GetDiffuseAlbedo :135-138
    const int bounce = (int)(c.kind & 0xFFu);
    const int dd = isRoot ? 0 : (int)((c.kind >> 8) & 0xFFu) + (mtl_is_photon_surface(m) ? 1 : 0);
    lg_shade(g, m, sx, sh.obj, isRoot ? 0u : slot, lg_absorption_factor(c.a, dist), bounce, dd, S);
    lg_unwind(S);
}

// every queued call traced to its end, one after the other (legacy_secondary_kernel); firstDist receives the `dist`
// result of the first call traced (the one a caller of rtx_trace_secondary asked for)
template <class G>
__device__ __forceinline__ void lg_run(const G& g, unsigned long long key, LgState& S, float* firstDist = nullptr) {
    bool first = true;
    while (S.sp > 0) {
        const LgCall& c = S.stack[--S.sp];
        const RayT r{c.o.x, c.o.y, c.o.z, c.d.x, c.d.y, c.d.z, RTX_BIG};
        if (!(c.kind & LG_SHADOW_CALL)) S.rays++;
        uint32_t nc = 0, tc = 0;
        const HitT h = traverse_closest<false>(g.sc, r, RTX_HIT_FRONT_AND_BACK, g.st, &nc, &tc);
        lg_complete(g, key, c, h, false, S, nullptr, first ? firstDist : nullptr);
        first = false;
    }
}

// TraceLegacy (RayCaster.cpp:579-631) for every camera sample of the pass, from the primary hits of extend_kernel.
// Persistent grid: thread t handles queue items t, t + threads, ... (the traversal spill slab is sized for the grid).
// Three resident waves per SIMD (168 VGPR): with the traversal state, the call in flight and the call stack live across
// the whole loop, more waves mean more scratch traffic in the traversal steps (4 spp frame at batch 48: 157 / 125 / 138 /
// 167 / 178 ms for 2 / 3 / 4 / 5 / 6 waves). [The earlier form of this kernel - one lane tracing each call to its end on
// its own - was latency-bound instead and ran best at six waves: 285 ms at the compiler's 2 waves, 195 at 6.]
#ifndef LEGACY_WAVES_PER_EU
#define LEGACY_WAVES_PER_EU 3
#endif
template <bool XF, bool PLAIN>
__attribute__((amdgpu_waves_per_eu(LEGACY_WAVES_PER_EU)))
__global__ __launch_bounds__(TRAV_BLOCK) void legacy_kernel(DevScene scArg, DevCamera cam, PathPool pool, DevPhotonMap pmGlobal,
                                                            DevPhotonMap pmCaustic, SegQ qIn, const uint32_t* __restrict__ queueIn,
                                                            uint32_t n, unsigned long long seed, int sBase, int S,
                                                            const int* __restrict__ pixels, uint2* spill, char* state) {
    __shared__ uint32_t qtab[SEGQ_TABLE_WORDS];
    __shared__ uint2 lds_stack[STACK_LDS * TRAV_BLOCK];
    segq_load_table(qIn, qtab);
    __syncthreads();
    const DevSceneT<XF> sc(scArg);
    const uint32_t gtid = blockIdx.x * blockDim.x + threadIdx.x, nThreads = gridDim.x * blockDim.x;
    const LgSceneT<DevSceneT<XF>, PLAIN> g{sc, pmGlobal, pmCaustic, make_trav_stack<true>(&lds_stack[threadIdx.x], spill, gtid, nThreads)};
    unsigned long long rays = 0, shadows = 0;
    // Every lane is a small state machine: it holds one camera sample and its stack of pending calls (secondary rays and
    // the shadow queries of the Shade() just evaluated), and at most one ray in flight. The wave runs ONE kind of step per
    // iteration, as extend_kernel does: an inner-node or a leaf step for the lanes whose ray is in flight - whichever
    // group is larger - or, once LG_BATCH lanes hold a finished ray (or nothing is in flight), the completion of those
    // rays: a shadow query adds its light's contribution if nothing blocks it; a secondary ray (or the primary hit, which
    // arrives traced by extend_kernel) is resolved, bump-mapped and shaded, which queues its own calls. A lane whose sample
    // is done takes its next one at once (one recursion tree can be 30 times another's).
    LgState st;
    st.L = mk3(0.0f);
    st.sp = st.fp = 0;
    st.rays = st.shadows = 0;
    st.topLocal = mk3(0.0f);
    st.topBase = st.topFlags = 0u;
    st.stack = reinterpret_cast<LgCall*>(state + (size_t)gtid * LG_LANE_BYTES);
    st.frames = reinterpret_cast<LgFrame*>(st.stack + LG_STACK);
    TravState ts;
    ts.cur = RTX_TRAV_DONE;
    ts.sp = 0;
    TravStack tstack = g.st;
    // the call being traced stays where it was popped from, st.stack[st.sp] (the slot above the stack's top), until its ray
    // has finished: only the ray itself (ts) is live across the traversal steps
    HitT rootH{0.0f, 0.0f, 0.0f, -1};
    bool have = false;            // this lane holds a sample
    bool inFlight = false;        // c is being traced (ts)
    bool rootPending = false;     // c is the primary hit, waiting for its Shade()
    bool finished = false;        // no samples left for this lane
    uint32_t slot = 0;
    float primaryDist = RTX_BIG;
    unsigned long long key = 0;
    uint32_t i = gtid;
    for (;;) {
        // ---- idle lanes take their next call, or their next sample ----
        if (!inFlight && !rootPending && !finished) {
            if (st.sp > 0) {
                const LgCall& cc = st.stack[--st.sp];
                trav_init(ts, sc, RayT{cc.o.x, cc.o.y, cc.o.z, cc.d.x, cc.d.y, cc.d.z, RTX_BIG});
                inFlight = true;
                if (!(cc.kind & LG_SHADOW_CALL)) st.rays++;
            } else {
                if (have) {
                    pool.L[slot] = make_float4(st.L.x, st.L.y, st.L.z, primaryDist);
                    rays += st.rays;
                    shadows += st.shadows;
                    have = false;
                }
                if (i >= n) {
                    finished = true;
                } else {
                    slot = queueIn[segq_item_index(qtab, i)];
                    i += nThreads;
                    const float4 ra = pool.rays[2 * (size_t)slot], rb = pool.rays[2 * (size_t)slot + 1];
                    const float4 hv = pool.hits[slot];
                    const RayT ray{ra.x, ra.y, ra.z, rb.x, rb.y, rb.z, ra.w};
                    const HitT h{hv.x, hv.y, hv.z, __float_as_int(hv.w)};
                    const uint32_t lane64 = slot & 63u, cs = slot >> 6;
                    const int pix = pixels[(cs / (uint32_t)S) * 64u + lane64];
                    const int sIdx = sBase + (int)(cs % (uint32_t)S);
                    const f3 rdir = mk3(ray.dx, ray.dy, ray.dz), rorg = mk3(ray.ox, ray.oy, ray.oz);
                    key = rng_key(seed, (uint32_t)pix, (uint32_t)sIdx);
                    st.L = mk3(0.0f);
                    st.rays = st.shadows = 0;
                    primaryDist = RTX_BIG;
                    have = true;
                    if (h.slot == -1) {   // :583-593
                        const f3 nd = normalized3(rdir);
                        float uEnv = 0.0f, vEnv = 0.0f;
                        dir_to_screen_uv(cam, nd, uEnv, vEnv);
                        const bool hasBackground = has_map_or_value(sc.bg, sc.bg_tex);
                        const bool hasEnvironment = has_map_or_value(sc.env, sc.env_tex);
                        const f3 backplate = hasBackground ? eval_tc(sc, sc.bg, sc.bg_tex, mk3(uEnv, vEnv, 0.0f)) : mk3(0.0f);
                        f3 envLight = mk3(0.0f);
                        if (hasEnvironment) envLight = sc.env_tex >= 0 ? ld3(sc.env) * tex_eval_env(sc, sc.env_tex, nd) : ld3(sc.env);
                        const f3 bgEmit = (hasBackground && sc.bg_emission > 0.0f) ? backplate * sc.bg_emission : mk3(0.0f);
                        st.L = (hasBackground ? backplate : envLight) + bgEmit;
                    } else if (h.slot < -1) {   // a renderable light in front of the geometry: :613-624
                        SurfHit sl;
                        resolve_light(sc.lights, ray, h, sl);
                        primaryDist = len3(sl.p - rorg);
                        const DevLight& light = sc.lights[sl.light];
                        st.L = ld3(light.intensity) / (PI_F * light.size * light.size);
                    } else {
                        rootPending = true;
                        rootH = h;
                        LgCall& cc = st.stack[0];   // st.sp == 0 here
                        cc.o = rorg;
                        cc.d = rdir;
                        cc.a = cc.b = mk3(0.0f);
                        cc.node = 0u;
                        cc.kind = 0u;
                    }
                }
            }
        }
        // ---- what the wave does this iteration ----
        const bool travDone = inFlight && ts.cur == RTX_TRAV_DONE;
        const bool complete = rootPending || travDone;
        const bool atInner = inFlight && !travDone && trav_is_inner(ts.cur);
        const bool atLeaf = inFlight && !travDone && !atInner;
        const uint32_t nComplete = (uint32_t)__popcll(__ballot(complete));
        const uint32_t nInner = (uint32_t)__popcll(__ballot(atInner)), nLeaf = (uint32_t)__popcll(__ballot(atLeaf));
        if (nComplete + nInner + nLeaf == 0u) {
            if (__ballot(!finished) == 0ull) break;
            continue;
        }
        if (nComplete < LG_BATCH && nInner + nLeaf > 0u) {
            if (nLeaf * LG_LEAF_WEIGHT >= nInner) {
                if (atLeaf) trav_leaf_step(ts, sc, tstack, RTX_HIT_FRONT_AND_BACK);
            } else if (atInner) {
                trav_inner_step(ts, sc, tstack);
            }
            continue;
        }
        if (!complete) continue;
        // ---- completion of a finished ray ----
        const bool isRoot = rootPending;
        if (!isRoot && !trav_verify(ts, sc)) continue;   // general kernels: culled by its object's box in the reference, traced again without it
        const HitT h = isRoot ? rootH : ts.best;
        inFlight = false;
        rootPending = false;
        LgRootOut ro;
        ro.hit = ro.shaded = false;
        const LgCall& c = st.stack[st.sp];   // read before lg_shade pushes over it
        lg_complete(g, key, c, h, isRoot, st, isRoot ? &ro : nullptr, nullptr);
        if (isRoot && ro.hit) {
            primaryDist = ro.primaryDist;
            pool.normal[slot] = make_float4(ro.normal.x, ro.normal.y, ro.normal.z, 1.0f);
            pool.albedo[slot] = ro.shaded ? make_float4(ro.albedo.x, ro.albedo.y, ro.albedo.z, 1.0f) : make_float4(0.0f, 0.0f, 0.0f, 1.0f);
        }
    }
    atomicAdd(&pool.rayCounters[0], shadows);
    atomicAdd(&pool.rayCounters[1], rays);
}


// ShadeInfoAdapter::TraceSecondaryRay (ShadeInfoAdapter.h:94-231) for n directions from one shading point: the entry
// point behind rtx_trace_secondary. out[i] = {colour, dist}.
template <bool XF>
__global__ __launch_bounds__(TRAV_BLOCK) void legacy_secondary_kernel(DevScene scArg, DevPhotonMap pmGlobal, DevPhotonMap pmCaustic,
                                                                      rtx_shade_point at, const float* __restrict__ dirs, uint32_t n,
                                                                      int recursive, int child, unsigned long long seed,
                                                                      float4* __restrict__ out, uint2* spill) {
    __shared__ uint2 lds_stack[STACK_LDS * TRAV_BLOCK];
    const DevSceneT<XF> sc(scArg);
    const uint32_t gtid = blockIdx.x * blockDim.x + threadIdx.x, nThreads = gridDim.x * blockDim.x;
    const LgSceneT<DevSceneT<XF>> g{sc, pmGlobal, pmCaustic, make_trav_stack(&lds_stack[threadIdx.x], spill, gtid, nThreads)};
    for (uint32_t i = gtid; i < n; i += nThreads) {
        LgCall stackMem[LG_STACK];     // a handful of rays per call of the entry point: compiler scratch will do
        LgFrame frameMem[LG_FRAMES];
        LgState st;
        st.stack = stackMem;
        st.frames = frameMem;
        st.L = mk3(0.0f);
        st.sp = st.fp = 0;
        st.rays = st.shadows = 0;
        st.topLocal = mk3(0.0f);
        st.topBase = st.topFlags = 0u;
        float dist = RTX_BIG;
        if (at.bounce < LG_MAX_BOUNCES) {   // CanBounce (:97)
            SampCtx sx;
            sx.P = ld3(at.P);
            sx.N = ld3(at.N);
            sx.V = sx.GN = sx.uvw = mk3(0.0f);
            sx.front = true;
            sx.key = rng_key(seed, (uint32_t)at.pixel, (uint32_t)at.sample);
            sx.stream = at.node;
            // recursive == false: the Shade() reached behaves as at the depth limit (:220-223)
            lg_push(st, sx, mk3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]), mk3(0.0f), at.node * 3u + (uint32_t)child,
                    recursive ? at.bounce + 1 : LG_MAX_BOUNCES, at.diffuse_depth, (uint32_t)child & 3u);
            lg_run(g, sx.key, st, &dist);
        }
        out[i] = make_float4(st.L.x, st.L.y, st.L.z, dist);
    }
}

size_t legacy_state_bytes(const LaunchCfg& cfg) { return (size_t)cfg.gridBlocks * TRAV_BLOCK * LG_LANE_BYTES; }

void launch_legacy_secondary(const DevScene& sc, const LaunchCfg& cfg, const DevPhotonMap& pmGlobal, const DevPhotonMap& pmCaustic,
                             const rtx_shade_point& at, const float* dirs, uint32_t n, int recursive, int child, unsigned long long seed,
                             float4* out) {
    if (!n) return;
    const uint32_t need = (n + TRAV_BLOCK - 1) / TRAV_BLOCK;
    const uint32_t grid = need < (uint32_t)cfg.gridBlocks ? need : (uint32_t)cfg.gridBlocks;
    if (sc.has_xf)
        hipLaunchKernelGGL(legacy_secondary_kernel<true>, dim3(grid), dim3(TRAV_BLOCK), 0, cfg.stream, sc, pmGlobal, pmCaustic, at, dirs, n,
                           recursive, child, seed, out, cfg.spill);
    else
        hipLaunchKernelGGL(legacy_secondary_kernel<false>, dim3(grid), dim3(TRAV_BLOCK), 0, cfg.stream, sc, pmGlobal, pmCaustic, at, dirs, n,
                           recursive, child, seed, out, cfg.spill);
}

void launch_legacy(const DevScene& sc, const LaunchCfg& cfg, const DevCamera& cam, const PathPool& pool, const DevPhotonMap& pmGlobal,
                   const DevPhotonMap& pmCaustic, const SegQ& qIn, const uint32_t* queueIn, uint32_t n, unsigned long long seed,
                   int sBase, int S, const int* pixels, int blinnOnly, void* state) {
    if (!n) return;
    const uint32_t need = (n + TRAV_BLOCK - 1) / TRAV_BLOCK;
    const uint32_t grid = need < (uint32_t)cfg.gridBlocks ? need : (uint32_t)cfg.gridBlocks;
    const bool plain = blinnOnly && pmGlobal.n <= 0 && pmCaustic.n <= 0;
#define RTX_LAUNCH_LEGACY(X, P)                                                                                                          \
    hipLaunchKernelGGL((legacy_kernel<X, P>), dim3(grid), dim3(TRAV_BLOCK), 0, cfg.stream, sc, cam, pool, pmGlobal, pmCaustic, qIn, queueIn, \
                       n, seed, sBase, S, pixels, cfg.spill, (char*)state)
    if (sc.has_xf) {
        if (plain) RTX_LAUNCH_LEGACY(true, true);
        else RTX_LAUNCH_LEGACY(true, false);
    } else {
        if (plain) RTX_LAUNCH_LEGACY(false, true);
        else RTX_LAUNCH_LEGACY(false, false);
    }
#undef RTX_LAUNCH_LEGACY
}
