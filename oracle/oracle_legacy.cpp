// TEST INFRASTRUCTURE ONLY (see oracle_math.h). The legacy shading mode (SHADING_USE_LEGACY=1) of the CPU restatement,
// recursive like the reference.
//
// Restates (paths inside the reference repository):
//   TraceLegacy                                        RayTracer/RayCaster.cpp:579-631
//   ShadeInfoAdapter::TraceShadowRay / TraceSecondaryRay / CanBounce / diffuse depth
//                                                      RayTracer/ShadeInfoAdapter.h:40-231
//   PointLight::Illuminate / IntersectRay (legacy)     RayTracer/Shading_Legacy.cpp:115-235
//   FresnelExact / ApplyAbsorption / SampleGlossy      RayTracer/Shading_Legacy.cpp:296-363
//   MtlBlinn::Shade / MtlPhong::Shade / MtlMicrofacet::Shade        :366-694
//   MtlBlinn/MtlPhong/MtlMicrofacet::GenerateSample (legacy lobes)  :697-897
//   PhotonMappingConfig: USE_PHOTON_MAP_DIRECT = false, USE_PHOTON_MAP_INDIRECT = true  RayTracer/PhotonMappingConfig.h:8-11
//
// Random numbers: the reference draws from one sequential per-thread stream; here every Shade() call is a node of the
// recursion tree (root 0, children 3*id+1 diffuse / +2 reflection / +3 refraction) and its draws are keyed
// (pixel, sample) x node id x draw slot with the counter RNG of oracle_math.h, like the path mode.
#include "oracle_shade.h"

namespace orc {

enum { LG_DIM_REFL = 32, LG_DIM_REFR = 34 };   // draw slots of SampleGlossy (reflection, refraction); lights use DIM_LIGHT0 + 2l
static const int kMaxSpecularBounces = 5;      // ShadeInfoAdapter.h:245

struct LegacyCtx {
    const Scene& scene;
    const Camera& camera;
    const PhotonMapper* mapper;
    PathCounters* pc;
    uint64_t seed;
    uint32_t pixelIdx, sampleIdx;
};

// ShadeInfo of one hit: SamplerInfo + the legacy state (bounce, diffuse depth, node) + the RNG node id
struct LegacyInfo {
    SInfo s;
    int obj = -1;
    int bounce = 0, diffuseDepth = 0;
    uint32_t node = 0;
};

static inline float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Shading_Legacy.cpp:76-82 (same construction as Shading.cpp:31-37)
static inline void build_onb(V3 n, V3& u, V3& v) {
    if (std::fabs(n.x) > 0.5f) u = cross(V3(0, 1, 0), n);
    else u = cross(V3(1, 0, 0), n);
    u = normalized(u);
    v = cross(n, u);
    v = normalized(v);
}

// Shading_Legacy.cpp:212-235
static bool light_intersect_legacy(const Light& L, const Ray& ray, Hit& h) {
    if (L.size <= 0.0f) return false;
    V3 toC = L.position - ray.p;
    float tca = dot(toC, ray.dir);
    if (tca < 0) return false;
    float d2 = len2(toC) - tca * tca;
    if (d2 > L.size * L.size) return false;
    float thc = std::sqrt(L.size * L.size - d2);
    float t0 = tca - thc;
    float t1 = tca + thc;
    if (t0 > t1) std::swap(t0, t1);
    if (t0 < 0) t0 = t1;
    if (t0 < 0) return false;
    h.z = t0;
    h.p = ray.p + ray.dir * t0;
    h.N = normalized(h.p - L.position);
    h.GN = h.N;
    h.mtlID = 0;
    h.light = true;
    return true;
}

// ShadeInfoAdapter.h:49-88
static float trace_shadow_ray(const LegacyCtx& c, const LegacyInfo& li, V3 dir, float t_max) {
    const float Llen = len(dir);
    if (Llen < 1e-12f) return 1.0f;
    const bool finiteSegment = (t_max < BIGFLOAT * 0.5f);
    const float segLenWorld = finiteSegment ? (Llen * t_max) : BIGFLOAT;
    const V3 Ldir = dir / Llen;
    const V3 N = li.s.N;
    const float nDotL = dot(N, Ldir);
    const float baseBias = 1e-4f;
    const float nBias = baseBias * (1.0f + 2.0f * (1.0f - std::fabs(nDotL)));
    const float dBias = baseBias * 0.5f;
    Ray sRay;
    sRay.p = li.s.P + ((nDotL >= 0.0f) ? N : -N) * nBias + Ldir * dBias;
    sRay.dir = Ldir;
    Hit h;
    h.init();
    h.z = BIGFLOAT;
    if (c.pc) c.pc->shadow++;
    if (c.scene.intersect_node(sRay, h, RTX_HIT_FRONT_AND_BACK, true)) {
        const float dist = len(h.p - sRay.p);
        if (h.obj == li.obj && dist < (nBias + dBias) * 6.0f) return 1.0f;
        if (!finiteSegment) return dist > 0.0f ? 0.0f : 1.0f;
        const float allowance = nBias * 1.5f;
        return (dist < segLenWorld - allowance) ? 0.0f : 1.0f;
    }
    return 1.0f;
}

// Shading_Legacy.cpp:115-210
static V3 illuminate(const LegacyCtx& c, const LegacyInfo& li, const Light& light, int lightIdx, V3& dir) {
    V3 P = li.s.P;
    V3 toC = light.position - P;
    float distC = len(toC);
    if (distC < 1e-20f) {
        dir = V3(0, 0, 0);
        return V3(0, 0, 0);
    }
    V3 L = toC / distC;
    dir = L;
    float atten = 1.0f;
    if (light.attenuation > 0.0f) atten = 1.0f / ((light.attenuation * distC) * (light.attenuation * distC));
    if (light.size <= 0.0f) return light.intensity * trace_shadow_ray(c, li, toC, 1.0f) * atten;
    V3 U, V;
    build_onb(L, U, V);
    const int kMinSamples = 1, kMaxSamples = 4;
    const float jx = li.s.rnd(DIM_LIGHT0 + 2 * lightIdx);
    const float jy = li.s.rnd(DIM_LIGHT0 + 2 * lightIdx + 1);
    int n = 0, lit = 0, dark = 0;
    auto sampleOnce = [&](int sIdx) {
        float h2 = halton(sIdx + 1, 2);
        h2 = h2 + jx;
        h2 -= std::floor(h2);
        float h3 = halton(sIdx + 1, 3);
        h3 = h3 + jy;
        h3 -= std::floor(h3);
        float gamma = rtx_asinf(light.size / distC);
        float u1 = h2, u2 = h3;
        float cosTheta = 1.0f - u1 * (1.0f - rtx_cosf(gamma));
        float sinTheta = std::sqrt(std::max(0.0f, 1.0f - cosTheta * cosTheta));
        float phi = 2.0f * PI_F * u2;
        V3 sampleDir = normalized(U * rtx_cosf(phi) * sinTheta + V * rtx_sinf(phi) * sinTheta + L * cosTheta);
        float vis = trace_shadow_ray(c, li, sampleDir * distC, 1.0f);
        ++n;
        if (vis >= 0.999f) ++lit;
        else if (vis <= 0.001f) ++dark;
        return vis;
    };
    float sumVis = 0.0f;
    for (int i = 0; i < kMinSamples; ++i) sumVis += sampleOnce(i);
    if (lit == kMinSamples || dark == kMinSamples) {
        float avgVis = sumVis / float(n);
        return light.intensity * avgVis * atten;
    }
    // not reached while TraceShadowRay returns exactly 0 or 1 (it does, ShadeInfoAdapter.h:75-87); kept as written
    for (int i = kMinSamples; i < kMaxSamples; ++i) {
        sumVis += sampleOnce(i);
        if (i >= kMinSamples + 2) {
            if (lit >= n * 0.75f || dark >= n * 0.75f) break;
        }
    }
    float avgVis = clampf(sumVis / float(n), 0.0f, 1.0f);
    return light.intensity * avgVis * atten;
}

// Shading_Legacy.cpp:296-318
static float fresnel_exact(V3 I, V3 N, float ior) {
    float cosi = clampf(dot(I, N), -1.0f, 1.0f);
    float etai = 1.0f, etat = ior;
    if (cosi > 0.0f) {
        std::swap(etai, etat);
        N = -N;
    }
    cosi = std::fabs(cosi);
    float sint = etai / etat * std::sqrt(std::max(0.0f, 1.0f - cosi * cosi));
    if (sint >= 1.0f) return 1.0f;
    float cost = std::sqrt(std::max(0.0f, 1.0f - sint * sint));
    float Rs = ((etat * cosi) - (etai * cost)) / ((etat * cosi) + (etai * cost));
    float Rp = ((etai * cosi) - (etat * cost)) / ((etai * cosi) + (etat * cost));
    return clampf(0.5f * (Rs * Rs + Rp * Rp), 0.0f, 1.0f);
}

// Shading_Legacy.cpp:321-329
static inline V3 apply_absorption_legacy(V3 c, V3 absorption, float dist) {
    if (dist <= 0.0f) return c;
    if (absorption.x == 0.0f && absorption.y == 0.0f && absorption.z == 0.0f) return c;
    const float ar = rtx_expf(-absorption.x * dist);
    const float ag = rtx_expf(-absorption.y * dist);
    const float ab = rtx_expf(-absorption.z * dist);
    return V3(c.x * ar, c.y * ag, c.z * ab);
}

static V3 shade(const LegacyCtx& c, const Material& m, const LegacyInfo& li);

// scene->environment.EvalEnvironment(dir) (ShadeInfoAdapter.h:228)
static V3 eval_environment(const Scene& scene, V3 nd) {
    return scene.envTex >= 0 ? scene.envColor * scene.texmaps[scene.envTex].eval_env(nd) : scene.envColor;
}

// ShadeInfoAdapter.h:94-231; `child` = 1 diffuse, 2 reflection, 3 refraction (RNG node of the Shade() it leads to).
// The isReflection flag selects between two branches (:115-183) with the same outcome: the nearest renderable light
// in front of the geometry hit wins, its radiance clamped to [0,1].
static V3 trace_secondary_ray(const LegacyCtx& c, const LegacyInfo& li, V3 dir, float& dist, int child, bool recursive = true) {
    dist = BIGFLOAT;
    if (!(li.bounce < kMaxSpecularBounces)) return V3(0, 0, 0);
    const float baseBias = 1e-4f;
    float dlen = len(dir);
    if (dlen < 1e-20f) return V3(0, 0, 0);
    dir = dir / dlen;
    const V3 N = li.s.N;
    const float nDotD = dot(N, dir);
    const float nBias = baseBias * (1.0f + 2.0f * (1.0f - std::fabs(nDotD)));
    const float dBias = baseBias * 0.5f;
    Ray ray;
    ray.p = li.s.P + ((nDotD >= 0.0f) ? N : -N) * nBias + dir * dBias;
    ray.dir = dir;
    if (c.pc) c.pc->bounce++;

    float bestLightT = BIGFLOAT;
    int visibleLight = -1;
    for (size_t k = 0; k < c.scene.lights.size(); ++k) {
        const Light& L = c.scene.lights[k];
        if (!L.renderable()) continue;
        Hit hl;
        hl.init();
        if (light_intersect_legacy(L, ray, hl)) {
            if (hl.z > 0.0f && hl.z < bestLightT) {
                bestLightT = hl.z;
                visibleLight = (int)k;
            }
        }
    }
    Hit h;
    h.init();
    h.z = BIGFLOAT;
    const bool hitGeom = c.scene.intersect_node(ray, h, RTX_HIT_FRONT_AND_BACK, true);
    if (visibleLight >= 0 && (!hitGeom || bestLightT < h.z)) {
        dist = bestLightT;
        V3 rad = light_radiance(c.scene.lights[visibleLight]);
        return V3(clampf(rad.x, 0.0f, 1.0f), clampf(rad.y, 0.0f, 1.0f), clampf(rad.z, 0.0f, 1.0f));
    }
    if (hitGeom) {
        const float hitDist = len(h.p - ray.p);
        dist = h.front ? hitDist : 0.0f;
        const Material* mtl = (h.obj >= 0 && c.scene.meshes[h.obj].material >= 0) ? &c.scene.materials[c.scene.meshes[h.obj].material] : nullptr;
        if (!mtl) return V3(0, 0, 0);
        apply_bump(c.scene, mtl, h);
        LegacyInfo s2;
        s2.s = li.s;   // keeps the RNG key
        s2.s.set_hit(ray, h);
        s2.obj = h.obj;
        s2.diffuseDepth = li.diffuseDepth + (mtl_is_photon_surface(*mtl) ? 1 : 0);
        s2.bounce = recursive ? li.bounce + 1 : kMaxSpecularBounces;   // :217-223; Shading_Legacy.cpp always recurses
        s2.node = li.node * 3u + (uint32_t)child;
        s2.s.stream = s2.node;
        return shade(c, *mtl, s2);
    }
    return eval_environment(c.scene, normalized(ray.dir));
}

// Shading_Legacy.cpp:332-363 with glossySamples = 1 (:474, :490)
static V3 sample_glossy(const LegacyCtx& c, const LegacyInfo& li, V3 axis, float alpha, bool isReflection, V3 V, V3 absorption) {
    V3 U, Vb;
    build_onb(axis, U, Vb);
    const int d0 = isReflection ? LG_DIM_REFL : LG_DIM_REFR;
    float u1 = li.s.rnd(d0), u2 = li.s.rnd(d0 + 1);
    float cosTheta = rtx_powf(u1, 1.0f / (alpha + 1.0f));
    float sinTheta = std::sqrt(std::max(0.0f, 1.0f - cosTheta * cosTheta));
    float phi = 2.0f * PI_F * u2;
    V3 H = normalized(U * rtx_cosf(phi) * sinTheta + Vb * rtx_sinf(phi) * sinTheta + axis * cosTheta);
    V3 dir = isReflection ? (2.0f * dot(V, H) * H - V) : H;
    dir = normalized(dir);
    float dist;
    V3 col = trace_secondary_ray(c, li, dir, dist, isReflection ? 2 : 3);
    if (!isReflection) col = apply_absorption_legacy(col, absorption, dist);
    return col / 1.0f;
}

// Shading_Legacy.cpp:697-788 (Blinn), :790-874 (Phong), :876-897 (microfacet)
static bool generate_sample_legacy(const LegacyCtx& c, const Material& m, const LegacyInfo& li, V3& dir, DirInfo& si) {
    const Scene& sc = c.scene;
    const SInfo& s = li.s;
    if (m.d.type == RTX_MTL_MICROFACET) {
        V3 N = normalized(s.N);
        V3 U, Vb;
        build_onb(N, U, Vb);
        float u1 = s.rnd(DIM_U1), u2 = s.rnd(DIM_U2);
        float cosTheta = std::sqrt(u1);
        float sinTheta = std::sqrt(1.0f - cosTheta * cosTheta);
        float phi = 2.0f * PI_F * u2;
        dir = U * (sinTheta * rtx_cosf(phi)) + Vb * (sinTheta * rtx_sinf(phi)) + N * cosTheta;
        si.mult = sc.eval_tc(m.d.diffuse, m.d.diffuse_tex, s.uvw);
        si.prob = cosTheta / PI_F;
        si.lobe = LOBE_DIFFUSE;
        return true;
    }
    const bool phong = m.d.type == RTX_MTL_PHONG;
    V3 N = normalized(s.N);
    if (!s.front) N = -N;
    V3 V = normalized(s.V);
    float u1 = s.rnd(DIM_U1), u2 = s.rnd(DIM_U2);
    V3 diff = sc.eval_tc(m.d.diffuse, m.d.diffuse_tex, s.uvw);
    V3 refl = sc.eval_tc(m.d.reflection, m.d.reflection_tex, s.uvw);
    V3 refr = sc.eval_tc(m.d.refraction, m.d.refraction_tex, s.uvw);
    float diffWeight = vsum(diff), reflWeight = vsum(refl), refrWeight = vsum(refr);
    float total = diffWeight + reflWeight + refrWeight;
    if (total < 1e-6f) return false;
    float pick = s.rnd(DIM_LOBE) * total;
    const float alpha = sc.eval_tf(m.d.glossiness, m.d.glossiness_tex, s.uvw);   // sInfo.Eval(Glossiness())
    if (pick < diffWeight) {
        V3 U, Vb;
        build_onb(N, U, Vb);
        float cosTheta = std::sqrt(u1);
        float sinTheta = std::sqrt(1.0f - cosTheta * cosTheta);
        float phi = 2.0f * PI_F * u2;
        dir = U * (sinTheta * rtx_cosf(phi)) + Vb * (sinTheta * rtx_sinf(phi)) + N * cosTheta;
        si.mult = diff;
        si.prob = diffWeight / total;
        si.lobe = LOBE_DIFFUSE;
    } else if (pick < diffWeight + reflWeight) {
        V3 R = 2.0f * dot(N, V) * N - V;
        R = normalized(R);
        V3 U, Vb;
        build_onb(R, U, Vb);
        float cosTheta = rtx_powf(u1, 1.0f / (alpha + 1.0f));
        float sinTheta = std::sqrt(std::max(0.0f, 1.0f - cosTheta * cosTheta));
        float phi = 2.0f * PI_F * u2;
        V3 H = U * (sinTheta * rtx_cosf(phi)) + Vb * (sinTheta * rtx_sinf(phi)) + R * cosTheta;
        H = normalized(H);
        if (phong) dir = H;                                   // :836-837
        else dir = normalized(2.0f * dot(V, H) * H - V);      // :744-748
        si.mult = refl;
        si.prob = reflWeight / total;
        si.lobe = LOBE_SPECULAR;
    } else {
        V3 I = -V;
        float n1 = 1.0f, n2 = m.d.ior;
        if (!s.front) std::swap(n1, n2);
        float eta = n1 / n2;
        float cosi = -dot(N, I);
        float k = 1.0f - eta * eta * (1.0f - cosi * cosi);
        if (k < 0.0f) return false;
        dir = eta * I + (eta * cosi - std::sqrt(k)) * N;
        dir = normalized(dir);
        V3 T = dir;
        if (alpha > 0.0f) {
            V3 U, Vb;
            build_onb(T, U, Vb);
            float cosTheta = rtx_powf(u1, 1.0f / (alpha + 1.0f));
            float sinTheta = std::sqrt(std::max(0.0f, 1.0f - cosTheta * cosTheta));
            float phi = 2.0f * PI_F * u2;
            dir = U * (sinTheta * rtx_cosf(phi)) + Vb * (sinTheta * rtx_sinf(phi)) + T * cosTheta;
            dir = normalized(dir);
        }
        si.mult = refr;
        si.prob = refrWeight / total;
        si.lobe = LOBE_TRANSMISSION;
    }
    return true;
}

static V3 photon_irradiance(const PhotonMap& pm, const LegacyInfo& li, float radius, int maxPhotons) {   // :32-51
    if (pm.ph.empty()) return V3(0, 0, 0);
    const int k = maxPhotons <= 30 ? 30 : (maxPhotons <= 50 ? 50 : 100);
    return pm.estimate(k, radius, li.s.P, li.s.N);
}

// MtlMicrofacet::Shade, Shading_Legacy.cpp:662-694
static V3 shade_microfacet(const LegacyCtx& c, const Material& m, const LegacyInfo& li) {
    const Scene& sc = c.scene;
    const V3 N = normalized(li.s.N);
    const V3 base = sc.eval_tc(m.d.diffuse, m.d.diffuse_tex, li.s.uvw);
    V3 result(0.0f);
    if (li.bounce == 0) {
        for (size_t i = 0; i < sc.lights.size(); ++i) {
            V3 Ldir;
            const V3 Li = illuminate(c, li, sc.lights[i], (int)i, Ldir);
            Ldir = normalized(Ldir);
            const float NdotL = std::max(0.0f, dot(N, Ldir));
            if (NdotL > 0.0f) result = result + base * Li * NdotL;
        }
    }
    V3 indirect(0.0f);
    if (vmax(base) > 1e-6f && li.bounce < kMaxSpecularBounces) {
        V3 indirDir;
        DirInfo si;
        if (generate_sample_legacy(c, m, li, indirDir, si)) {
            float dist;
            V3 Li = trace_secondary_ray(c, li, indirDir, dist, 1);
            indirect = Li * si.mult;
        }
    }
    result = result + indirect;
    result = result + sc.eval_tc(m.d.emission, m.d.emission_tex, li.s.uvw);
    return result;
}

// MtlBlinn::Shade :366-514, MtlPhong::Shade :517-659
static V3 shade(const LegacyCtx& c, const Material& m, const LegacyInfo& li) {
    if (m.d.type == RTX_MTL_MICROFACET) return shade_microfacet(c, m, li);
    const Scene& sc = c.scene;
    const SInfo& s = li.s;
    const bool phong = m.d.type == RTX_MTL_PHONG;
    V3 N = normalized(s.N);
    if (!s.front) N = -N;
    const V3 V = normalized(s.V);
    const V3 I = -V;
    const float alpha = sc.eval_tf(m.d.glossiness, m.d.glossiness_tex, s.uvw);   // sInfo.Eval(Glossiness())
    const V3 kd = sc.eval_tc(m.d.diffuse, m.d.diffuse_tex, s.uvw);
    const V3 ks = sc.eval_tc(m.d.specular, m.d.specular_tex, s.uvw);

    // direct light (USE_PHOTON_MAP_DIRECT = false: the :402-417 / :550-566 branch)
    V3 local(0.0f);
    for (size_t i = 0; i < sc.lights.size(); ++i) {
        V3 Ldir;
        const V3 Li = illuminate(c, li, sc.lights[i], (int)i, Ldir);
        Ldir = normalized(Ldir);
        const float NdotL = std::max(0.0f, dot(N, Ldir));
        if (!(NdotL > 0.0f)) continue;
        if (!phong) {
            local = local + (kd / PI_F) * Li * NdotL;
            const V3 H = normalized(Ldir + V);
            const float NdotH = std::max(0.0f, dot(N, H));
            if (NdotH > 0.0f) local = local + (ks * (alpha + 2.0f) / (8.0f * PI_F)) * Li * rtx_powf(NdotH, alpha);
        } else {
            V3 lightContrib = (kd / PI_F) * Li * NdotL;
            const V3 R = normalized(2.0f * dot(N, Ldir) * N - Ldir);
            const float RdotV = std::max(0.0f, dot(R, V));
            if (RdotV > 0.0f) lightContrib = lightContrib + (ks * (alpha + 1.0f) / (2.0f * PI_F)) * Li * rtx_powf(RdotV, alpha);
            local = local + lightContrib;
        }
    }

    // indirect light :419-457
    V3 indirect(0.0f), causticTerm(0.0f);
    const bool useIndirectMap = c.mapper && !c.mapper->global.ph.empty();
    const bool hasDiffuse = vmax(kd) > 1e-6f;
    if (hasDiffuse && c.mapper && !c.mapper->caustic.ph.empty())
        causticTerm = (kd / PI_F) * photon_irradiance(c.mapper->caustic, li, c.mapper->causticRadius, c.mapper->searchCount);
    const bool canBounce = li.bounce < kMaxSpecularBounces;
    if (useIndirectMap && !(li.diffuseDepth <= 1)) {
        indirect = indirect + (kd / PI_F) * photon_irradiance(c.mapper->global, li, c.mapper->photonRadius, c.mapper->searchCount);
    } else if (hasDiffuse && canBounce) {
        V3 indirDir;
        DirInfo si;
        if (generate_sample_legacy(c, m, li, indirDir, si)) {
            float dist;
            V3 Li = trace_secondary_ray(c, li, indirDir, dist, 1);
            indirect = Li * si.mult / si.prob;
        }
    }
    indirect = indirect + causticTerm;
    local = local + indirect;

    // reflection / refraction :459-508
    const V3 refl = sc.eval_tc(m.d.reflection, m.d.reflection_tex, s.uvw);
    const V3 refrRaw = sc.eval_tc(m.d.refraction, m.d.refraction_tex, s.uvw);
    const float ior = m.d.ior;
    const bool isDielectric = vmax(refrRaw) > 1e-6f;
    const V3 refrTint = isDielectric ? refrRaw : V3(0, 0, 0);
    V3 result = local;
    if ((isDielectric || vmax(refl) > 1e-4f) && canBounce) {
        const float kr = isDielectric ? fresnel_exact(I, N, ior) : 1.0f;
        V3 reflectionColor = sample_glossy(c, li, N, alpha, true, V, V3(0, 0, 0));
        V3 refractionColor(0.0f);
        if (isDielectric) {
            float n1 = 1.0f, n2 = ior;
            float cosi = -dot(N, I);
            if (!s.front) std::swap(n1, n2);
            const float eta = n1 / n2;
            const float k = 1.0f - eta * eta * (1.0f - cosi * cosi);
            if (k >= 0.0f) {
                V3 T = eta * I + (eta * cosi - std::sqrt(k)) * N;
                T = normalized(T);
                refractionColor = sample_glossy(c, li, T, alpha, false, V, V3(m.d.absorption[0], m.d.absorption[1], m.d.absorption[2]));
            }
        }
        if (isDielectric) {
            const V3 oneMinusRefl = V3(1.0f) - refl;
            result = local * oneMinusRefl + reflectionColor * kr + refractionColor * ((1.0f - kr) * refrTint);
        } else {
            result = local + reflectionColor * refl;
        }
    }
    result = result + sc.eval_tc(m.d.emission, m.d.emission_tex, s.uvw);
    return result;
}

// ShadeInfoAdapter::TraceSecondaryRay called from outside Shade() (the boundary entry point of SURVEY.md §8b)
V3 legacy_secondary(const Scene& scene, const PhotonMapper* mapper, const rtx_shade_point& at, V3 dir, bool recursive, int child,
                    uint64_t seed, float& dist) {
    Camera cam{};
    LegacyCtx c{scene, cam, mapper, nullptr, seed, (uint32_t)at.pixel, (uint32_t)at.sample};
    LegacyInfo li;
    li.s.seed = seed;
    li.s.ka = (uint32_t)at.pixel;
    li.s.kb = (uint32_t)at.sample;
    li.s.stream = at.node;
    li.s.P = V3(at.P[0], at.P[1], at.P[2]);
    li.s.N = V3(at.N[0], at.N[1], at.N[2]);
    li.obj = at.obj;
    li.bounce = at.bounce;
    li.diffuseDepth = at.diffuse_depth;
    li.node = at.node;
    return trace_secondary_ray(c, li, dir, dist, child, recursive);
}

// RayCaster.cpp:579-631
PathResult trace_legacy(const Scene& scene, const Camera& camera, Ray ray, int px, int py, int sampleIdx,
                        const PhotonMapper* mapper, uint64_t seed, PathCounters* pc) {
    PathResult res{V3(0, 0, 0), BIGFLOAT, V3(0, 0, 0), V3(0, 0, 0), false};
    LegacyCtx c{scene, camera, mapper, pc, seed, (uint32_t)(py * camera.W + px), (uint32_t)sampleIdx};
    Hit h;
    h.init();
    if (pc) pc->primary++;
    if (!scene.intersect_with_lights(ray, h, RTX_HIT_FRONT_AND_BACK)) {
        res.radiance = miss_radiance(scene, camera, normalized(ray.dir), true);
        return res;
    }
    res.primaryDist = len(h.p - ray.p);
    const Material* mtl = (h.obj >= 0 && scene.meshes[h.obj].material >= 0) ? &scene.materials[scene.meshes[h.obj].material] : nullptr;
    apply_bump(scene, mtl, h);
    LegacyInfo li;
    li.s.seed = seed;
    li.s.ka = c.pixelIdx;
    li.s.kb = c.sampleIdx;
    li.s.stream = 0;
    li.s.set_hit(ray, h);
    li.obj = h.obj;
    if (h.light && h.lightIdx >= 0) {
        res.radiance = light_radiance(scene.lights[h.lightIdx]);
        return res;
    }
    res.radiance = mtl ? shade(c, *mtl, li) : V3(0, 0, 0);
    res.albedo = mtl ? mtl_diffuse_albedo(scene, *mtl, li.s) : V3(0, 0, 0);
    res.normal = safe_normalize(h.GN);
    res.hit = true;
    return res;
}

}  // namespace orc
