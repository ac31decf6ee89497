// TEST INFRASTRUCTURE ONLY (see oracle_math.h). Light transport of the CPU restatement.
#include "oracle_shade.h"
#include <thread>
#include <cstring>

namespace orc {

Camera camera_from(const rtx_camera* c) {
    Camera k;
    k.pos = V3(c->pos[0], c->pos[1], c->pos[2]);
    k.dir = V3(c->dir[0], c->dir[1], c->dir[2]);
    k.up = V3(c->up[0], c->up[1], c->up[2]);
    k.fov = c->fov;
    k.focaldist = c->focaldist;
    k.dof = c->dof;
    k.W = c->width;
    k.H = c->height;
    k.srgb = c->srgb != 0;
    return k;
}

// ---- sampler --------------------------------------------------------------------------------------------
// framework Halton(index, base) (rng.h, absent): fp32 radical inverse, digit by digit.
float halton(int index, int base) {
    float r = 0;
    float f = 1.0f / (float)base;
    for (int i = index; i > 0; i /= base) {
        r += f * (float)(i % base);
        f /= (float)base;
    }
    return r;
}

// RayCaster.cpp:65-70
static inline uint64_t splitmix64(uint64_t& x) {
    uint64_t z = (x += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
// RayCaster.cpp:73-81
void pixel_shifts(int x, int y, float& shiftX, float& shiftY) {
    uint64_t seed = (uint64_t(uint32_t(x)) << 32) | uint64_t(uint32_t(y));
    uint64_t s = seed;
    uint64_t h1 = splitmix64(s);
    uint64_t h2 = splitmix64(s);
    shiftX = float((h1 & 0xFFFFFFFFULL) / double(0x100000000ULL));
    shiftY = float((h2 & 0xFFFFFFFFULL) / double(0x100000000ULL));
}

// RayCaster.cpp:717-762 (GenerateRay lambda + its captured frustum)
Ray generate_ray(const Camera& cam, float samplePosX, float samplePosY, int sampleIndex) {
    const int W = cam.W, H = cam.H;
    const float fovRad = cam.fov * 3.14159265f / 180.0f;
    const float halfY = rtx_tanf(0.5f * fovRad);
    const float aspect = (H > 0) ? float(W) / float(H) : 1.0f;
    const float halfX = halfY * aspect;
    const V3 f = cam.dir;
    const V3 r = normalized(cross(f, cam.up));
    const V3 u = normalized(cross(r, f));

    const float ndcY = 1.0f - (samplePosY / float(H)) * 2.0f;
    const float sy = ndcY * halfY;
    const float ndcX = (samplePosX / float(W)) * 2.0f - 1.0f;
    const float sx = ndcX * halfX;
    V3 dirVec = r * sx + u * sy + f;
    Ray ray;
    if (cam.dof > 1e-6f) {
        float focus = cam.focaldist;
        float aperture = cam.dof;
        V3 pixelWorld = cam.pos + dirVec * focus;
        float hx = halton(sampleIndex, 5);
        float hy = halton(sampleIndex, 7);
        float rd = std::sqrt(hx);
        float theta = 2.0f * PI_F * hy;
        float dx = rd * rtx_cosf(theta);
        float dy = rd * rtx_sinf(theta);
        V3 lensPoint = cam.pos + r * (dx * aperture) + u * (dy * aperture);
        ray.p = lensPoint;
        ray.dir = normalized(pixelWorld - lensPoint);
    } else {
        ray.p = cam.pos;
        ray.dir = normalized(dirVec);
    }
    return ray;
}

// CameraAdapter.h:7-27
Ray generate_camera_ray(const Camera& cam, int x, int y) {
    float fovRad = cam.fov * 3.14159265f / 180.0f;
    float halfY = rtx_tanf(0.5f * fovRad);
    float aspect = (cam.H > 0) ? float(cam.W) / float(cam.H) : 1.0f;
    float halfX = halfY * aspect;
    float ndcX = ((x + 0.5f) / float(cam.W)) * 2.0f - 1.0f;
    float ndcY = 1.0f - ((y + 0.5f) / float(cam.H)) * 2.0f;
    float sx = ndcX * halfX;
    float sy = ndcY * halfY;
    V3 f = cam.dir;
    V3 r = normalized(cross(f, cam.up));
    V3 u = normalized(cross(r, f));
    V3 dir = r * sx + u * sy + f;
    dir = normalized(dir);
    return Ray{cam.pos, dir};
}

// RayCaster.cpp:84-92, 643-690
float filter_radius(int type, float radiusOverride) {
    if (radiusOverride > 0.0f) return radiusOverride;
    switch (type) {
        case RTX_FILTER_TENT: return 1.0f;
        case RTX_FILTER_GAUSSIAN: return 1.5f;
        case RTX_FILTER_MITCHELL: return 2.0f;
        case RTX_FILTER_LANCZOS: return 2.0f;
    }
    return 1.0f;
}
float filter1d(int type, float radiusOverride, float x) {
    float radius = filter_radius(type, radiusOverride);
    switch (type) {
        case RTX_FILTER_GAUSSIAN: {
            float sigma = radius * 0.5f;
            if (sigma <= 0.0f) sigma = 0.5f;
            float t = x / sigma;
            return rtx_expf(-0.5f * t * t);
        }
        case RTX_FILTER_MITCHELL: {
            const float B = 1.0f / 3.0f;
            const float C = 1.0f / 3.0f;
            x = std::fabs(x);
            if (x < 1.0f)
                return ((12.0f - 9.0f * B - 6.0f * C) * x * x * x + (-18.0f + 12.0f * B + 6.0f * C) * x * x + (6.0f - 2.0f * B)) / 6.0f;
            else if (x < 2.0f)
                return ((-B - 6.0f * C) * x * x * x + (6.0f * B + 30.0f * C) * x * x + (-12.0f * B - 48.0f * C) * x + (8.0f * B + 24.0f * C)) / 6.0f;
            return 0.0f;
        }
        case RTX_FILTER_LANCZOS: {
            float a = radius;
            x = std::fabs(x);
            if (x >= a) return 0.0f;
            if (x < 1e-6f) return 1.0f;
            float pix = PI_F * x;
            return (a * rtx_sinf(pix) * rtx_sinf(pix / a)) / (pix * pix);
        }
        default: {
            x = std::fabs(x);
            if (x >= radius) return 0.0f;
            return 1.0f - x / radius;
        }
    }
}

// RayCaster.cpp:33-48; cy::Color::Linear2sRGB (absent): c<0.0031308 ? 12.92c : 1.055 c^0.41666 - 0.055
void to_color24(V3 c, bool srgb, unsigned char out[3]) {
    float v[3] = {c.x, c.y, c.z};
    for (int i = 0; i < 3; ++i) {
        float x = v[i];
        if (srgb) x = x < 0.0031308f ? x * 12.92f : rtx_powf(x, 0.41666f) * 1.055f - 0.055f;
        x = std::max(x, 0.0f);
        x = std::min(x, 1.0f);
        out[i] = (unsigned char)(x * 255.0f);
    }
}

// ---- lights ---------------------------------------------------------------------------------------------
// Light::Radiance for a sphere light (lights.h absent): intensity / (pi r^2), as GenerateSample uses :158.
V3 light_radiance(const Light& L) { return L.intensity / (PI_F * L.size * L.size); }

// Shading.cpp:118-165
bool light_generate_sample(const Light& L, const SInfo& sInfo, int lightIdx, V3& dir, DirInfo& si) {
    si.set_void();
    V3 P = sInfo.P;
    V3 toC = L.position - P;
    float dc = len(toC);
    if (dc < 1e-6f) return false;
    if (L.size <= 0.0f) {
        dir = toC / dc;
        float att_factor = (L.attenuation == 0.0f) ? 1.0f : L.attenuation * L.attenuation;
        si.mult = L.intensity / (att_factor * dc * dc);
        si.prob = 1.0f;
        si.dist = dc;
        si.lobe = LOBE_ALL;
        return true;
    }
    float u1 = sInfo.rnd(DIM_LIGHT0 + 2 * lightIdx);
    float u2 = sInfo.rnd(DIM_LIGHT0 + 2 * lightIdx + 1);
    float z = 1.0f - 2.0f * u1;
    float rxy = std::sqrt(std::max(0.0f, 1.0f - z * z));
    float phi = 2.0f * PI_F * u2;
    V3 onS = V3(rxy * rtx_cosf(phi), rxy * rtx_sinf(phi), z);
    V3 Lpos = L.position + L.size * onS;
    V3 wi = Lpos - P;
    float dist = len(wi);
    if (dist < 1e-6f) return false;
    dir = wi / dist;
    V3 nL = onS;
    float cosLight = std::max(0.0f, dot(-nL, dir));
    if (cosLight <= 0.0f) return false;
    float area = 4.0f * PI_F * L.size * L.size;
    float pdfArea = 2.0f / area;
    V3 radiance = L.intensity / (PI_F * L.size * L.size);
    float att_factor = (L.attenuation == 0.0f) ? 1.0f : L.attenuation * L.attenuation;
    si.mult = radiance / att_factor;
    si.prob = pdfArea;
    si.dist = dist;
    si.lobe = LOBE_ALL;
    return si.prob > 0.0f;
}

// Shading.cpp:167-209
void light_get_sample_info(const Light& L, V3 P, V3 dir, DirInfo& si) {
    si.set_void();
    V3 d = safe_normalize(dir);
    if (L.size <= 0.0f) {
        V3 toL = L.position - P;
        float dist = len(toL);
        if (dist < 1e-6f) return;
        if (dot(safe_normalize(toL), d) < 1.0f - 1e-4f) return;
        float att_factor = (L.attenuation == 0.0f) ? 1.0f : L.attenuation * L.attenuation;
        si.mult = L.intensity / (att_factor * dist * dist);
        si.prob = 1.0f;
        si.dist = dist;
        si.lobe = LOBE_ALL;
        return;
    }
    V3 oc = P - L.position;
    float b = dot(oc, d);
    float c = len2(oc) - L.size * L.size;
    float disc = b * b - c;
    if (disc < 0.0f) return;
    float t = -b - std::sqrt(std::max(0.0f, disc));
    if (t <= 1e-6f) return;
    V3 hit = P + d * t;
    V3 nL = safe_normalize(hit - L.position);
    float cosLight = std::max(0.0f, dot(-nL, d));
    if (cosLight <= 0.0f) return;
    float area = 4.0f * PI_F * L.size * L.size;
    float pdfArea = 2.0f / area;
    V3 radiance = L.intensity / (PI_F * L.size * L.size);
    float att_factor = (L.attenuation == 0.0f) ? 1.0f : L.attenuation * L.attenuation;
    si.mult = radiance / att_factor;
    si.prob = pdfArea;
    si.dist = t;
    si.lobe = LOBE_ALL;
}

// Shading.cpp:88-116; u[0..3] are the four RandomFloat() draws in call order
void light_random_photon(const Light& L, const float u[4], Ray& r, V3& c) {
    V3 pos = L.position;
    int k = 0;
    if (L.size > 0.0f) {
        float u1 = u[k++];
        float u2 = u[k++];
        float z = 1.0f - 2.0f * u1;
        float rxy = std::sqrt(std::max(0.0f, 1.0f - z * z));
        float phi = 2.0f * PI_F * u2;
        pos = L.position + L.size * V3(rxy * rtx_cosf(phi), rxy * rtx_sinf(phi), z);
    }
    float u1 = u[k++];
    float u2 = u[k++];
    float z = 1.0f - 2.0f * u1;
    float rxy = std::sqrt(std::max(0.0f, 1.0f - z * z));
    float phi = 2.0f * PI_F * u2;
    V3 dir(rxy * rtx_cosf(phi), rxy * rtx_sinf(phi), z);
    if (L.size > 0.0f) {
        V3 n = safe_normalize(pos - L.position);
        if (dot(dir, n) < 0.0f) dir = -dir;
    }
    r.p = pos;
    r.dir = dir;
    c = L.intensity * 4.0f * PI_F;
}

// ---- Phong / Blinn --------------------------------------------------------------------------------------
static inline void build_onb(V3 n, V3& u, V3& v) {  // Shading.cpp:31-37
    if (std::fabs(n.x) > 0.5f) u = cross(V3(0, 1, 0), n);
    else u = cross(V3(1, 0, 0), n);
    u = normalized(u);
    v = cross(n, u);
    v = normalized(v);
}
static inline V3 reflect(V3 w, V3 n) { return w - 2.0f * dot(w, n) * n; }  // :39-43
static inline float transmission_scale(float eta) { return std::min(eta * eta, 10.0f); }  // :45-48
static inline float fresnel_dielectric(float cosThetaI, float etaI, float etaT) {  // :50-61
    cosThetaI = clamp01(std::fabs(cosThetaI));
    float eta = etaI / etaT;
    float sin2T = eta * eta * (1.0f - cosThetaI * cosThetaI);
    if (sin2T >= 1.0f) return 1.0f;
    float cosThetaT = std::sqrt(std::max(0.0f, 1.0f - sin2T));
    float rParl = ((etaT * cosThetaI) - (etaI * cosThetaT)) / ((etaT * cosThetaI) + (etaI * cosThetaT));
    float rPerp = ((etaI * cosThetaI) - (etaT * cosThetaT)) / ((etaI * cosThetaI) + (etaT * cosThetaT));
    return clamp01(0.5f * (rParl * rParl + rPerp * rPerp));
}

struct PhongCtx {
    V3 N, V, R, kd, ks, kr, kt;
    float alpha, ior;
};

// Shading.cpp:227-278
static PhongCtx build_ctx(const Scene& sc, const rtx_material_desc& m, const SInfo& s) {
    PhongCtx c;
    c.N = normalized(s.N);
    if (!s.front) c.N = -c.N;
    c.V = safe_normalize(s.V);
    c.R = safe_normalize(reflect(-c.V, c.N));
    c.kd = sc.eval_tc(m.diffuse, m.diffuse_tex, s.uvw);
    c.ks = sc.eval_tc(m.specular, m.specular_tex, s.uvw);
    c.kr = sc.eval_tc(m.reflection, m.reflection_tex, s.uvw);
    c.kt = sc.eval_tc(m.refraction, m.refraction_tex, s.uvw);
    c.alpha = std::max(0.001f, sc.eval_tf(m.glossiness, m.glossiness_tex, s.uvw));   // Shading.cpp:238
    c.ior = m.ior;
    c.kd = clamp_max(c.kd, 1.0f);
    c.ks = clamp_max(c.ks, 1.0f);
    c.kr = clamp_max(c.kr, 1.0f);
    c.kt = clamp_max(c.kt, 1.0f);
    float krMax = vmax(c.kr);
    float ktMax = vmax(c.kt);
    if (krMax <= 0.0f && ktMax <= 0.0f) {
        c.ks = clamp_max(c.ks, 0.08f);
    }
    float reserve = std::min(1.0f, krMax + ktMax);
    float available = std::max(0.0f, 1.0f - reserve);
    V3 kdks = c.kd + c.ks;
    float kdksMax = vmax(kdks);
    if (kdksMax > 0.0f && kdksMax > available) {
        float scale = available / kdksMax;
        c.kd = c.kd * scale;
        c.ks = c.ks * scale;
    }
    return c;
}

// Shading.cpp:280-531
static bool sample_phong_blinn(const PhongCtx& ctx, bool phong, const SInfo& sInfo, V3& dir, DirInfo& si) {
    si.set_void();
    float etaI = sInfo.front ? 1.0f : ctx.ior;
    float etaT = sInfo.front ? ctx.ior : 1.0f;
    float F = (ctx.ior > 0.0f && vmax(ctx.kt) > 0.0f) ? fresnel_dielectric(std::fabs(dot(ctx.N, ctx.V)), etaI, etaT) : 0.0f;
    V3 fresnelRef = ctx.kt * F;
    V3 transColor = ctx.kt * (1.0f - F);
    V3 specColor = ctx.ks;
    if (vmax(ctx.kr) > 0.0f) specColor = V3(0.0f);
    specColor = clamp_max(specColor, 1.0f);

    float pd = luma1(ctx.kd);
    float ps = luma1(specColor);
    float pr = luma1(ctx.kr + fresnelRef);
    float pt = luma1(transColor);
    float total = pd + ps + pr + pt;
    if (total <= 0.0f) return false;

    float r = sInfo.rnd(DIM_LOBE) * total;
    const float a = ctx.alpha;

    if (r < pd) {
        V3 U, Vb;
        build_onb(ctx.N, U, Vb);
        float u1 = sInfo.rnd(DIM_U1);
        float u2 = sInfo.rnd(DIM_U2);
        float rxy = std::sqrt(u1);
        float phi = 2.0f * PI_F * u2;
        float x = rxy * rtx_cosf(phi);
        float y = rxy * rtx_sinf(phi);
        float z = std::sqrt(std::max(0.0f, 1.0f - u1));
        dir = safe_normalize(U * x + Vb * y + ctx.N * z);
        float cosN = std::max(0.0f, dot(ctx.N, dir));
        float pdf = cosN / PI_F;
        si.mult = ctx.kd * (1.0f / PI_F) * cosN;
        si.prob = (pd / total) * pdf;
        si.dist = 0.0f;
        si.lobe = LOBE_DIFFUSE;
        return si.prob > 0.0f && vmax(si.mult) > 0.0f;
    } else if (r < pd + ps || r < pd + ps + pr) {
        const bool specLobe = (r < pd + ps);
        V3 color = specColor;
        float pl = ps;
        if (!specLobe) {
            color = clamp_max(ctx.kr + fresnelRef, 1.0f);
            pl = pr;
        }
        if (phong) {
            V3 U, Vb;
            build_onb(ctx.R, U, Vb);
            float u1 = sInfo.rnd(DIM_U1);
            float u2 = sInfo.rnd(DIM_U2);
            float phi = 2.0f * PI_F * u1;
            float cosT = rtx_powf(u2, 1.0f / (a + 1.0f));
            float sinT = std::sqrt(std::max(0.0f, 1.0f - cosT * cosT));
            dir = safe_normalize(U * (rtx_cosf(phi) * sinT) + Vb * (rtx_sinf(phi) * sinT) + ctx.R * cosT);
            float cosR = std::max(0.0f, dot(ctx.R, dir));
            float cosN = std::max(0.0f, dot(ctx.N, dir));
            float pdfDir = ((a + 1.0f) / (2.0f * PI_F)) * rtx_powf(cosR, a);
            if (specLobe) si.mult = color * ((a + 2.0f) / (2.0f * PI_F)) * rtx_powf(cosR, a) * cosN;  // :342
            else {
                float brdf = ((a + 2.0f) / (2.0f * PI_F)) * rtx_powf(cosR, a);  // :398-399
                si.mult = color * brdf;
            }
            si.prob = (pl / total) * pdfDir;
            si.dist = 0.0f;
            si.lobe = LOBE_SPECULAR;
            return si.prob > 0.0f && vmax(si.mult) > 0.0f;
        } else {
            V3 U, Vb;
            build_onb(ctx.N, U, Vb);
            float u1 = sInfo.rnd(DIM_U1);
            float u2 = sInfo.rnd(DIM_U2);
            float phi = 2.0f * PI_F * u1;
            float cosT = rtx_powf(u2, 1.0f / (a + 1.0f));
            float sinT = std::sqrt(std::max(0.0f, 1.0f - cosT * cosT));
            V3 h = safe_normalize(U * (rtx_cosf(phi) * sinT) + Vb * (rtx_sinf(phi) * sinT) + ctx.N * cosT);
            float VoH;
            if (specLobe) {
                dir = safe_normalize(reflect(-ctx.V, h));  // :358
                VoH = std::max(0.0f, dot(ctx.V, h));       // :362
            } else {
                VoH = dot(ctx.V, h);                       // :417
                dir = safe_normalize(2.0f * VoH * h - ctx.V);
            }
            float NoL = std::max(0.0f, dot(ctx.N, dir));
            float NoH = std::max(0.0f, dot(ctx.N, h));
            if (NoL <= 0.0f || NoH <= 0.0f || VoH <= 1e-6f) return false;
            float pdfH = ((a + 1.0f) / (2.0f * PI_F)) * rtx_powf(NoH, a);
            float pdfDir = pdfH / (4.0f * std::max(1e-6f, VoH));
            float invVoH = 1.0f / std::max(1e-6f, VoH);
            V3 fSpec = color * ((a + 8.0f) / (8.0f * PI_F)) * rtx_powf(NoH, a) * invVoH;
            si.mult = fSpec;
            si.prob = (pl / total) * pdfDir;
            si.dist = 0.0f;
            si.lobe = LOBE_SPECULAR;
            return si.prob > 0.0f && vmax(si.mult) > 0.0f;
        }
    } else {
        float eta = sInfo.front ? 1.0f / ctx.ior : ctx.ior;
        V3 I = -ctx.V;
        if (phong) {
            float cosI = -dot(ctx.N, I);
            float k = 1.0f - eta * eta * (1.0f - cosI * cosI);
            if (k < 0.0f) {
                dir = safe_normalize(reflect(-ctx.V, ctx.N));
                si.mult = transColor;
                si.prob = (pt / total);
                si.dist = -1.0f;
                si.lobe = LOBE_SPECULAR;
                return si.prob > 0.0f && vmax(si.mult) > 0.0f;
            }
            float cosTIdeal = std::sqrt(k);
            V3 T = safe_normalize(eta * I + (eta * cosI - cosTIdeal) * ctx.N);
            V3 U, Vb;
            build_onb(T, U, Vb);
            float u1 = sInfo.rnd(DIM_U1);
            float u2 = sInfo.rnd(DIM_U2);
            float phi = 2.0f * PI_F * u1;
            float cosTheta = rtx_powf(u2, 1.0f / (a + 1.0f));
            float sinTheta = std::sqrt(std::max(0.0f, 1.0f - cosTheta * cosTheta));
            dir = safe_normalize(U * (rtx_cosf(phi) * sinTheta) + Vb * (rtx_sinf(phi) * sinTheta) + T * cosTheta);
            float cosT = std::max(0.0f, dot(T, dir));
            float absCosN = std::fabs(dot(ctx.N, dir));
            if (cosT <= 0.0f || absCosN <= 0.0f) return false;
            float pdfDir = ((a + 1.0f) / (2.0f * PI_F)) * rtx_powf(cosT, a);
            float etaScale = transmission_scale(eta);
            V3 fTrans = transColor * ((a + 2.0f) / (2.0f * PI_F)) * rtx_powf(cosT, a);
            si.mult = fTrans * etaScale;
            si.prob = (pt / total) * pdfDir;
            si.dist = 0.0f;
            si.lobe = LOBE_TRANSMISSION;
            return si.prob > 0.0f && vmax(si.mult) > 0.0f;
        } else {
            V3 U, Vb;
            build_onb(ctx.N, U, Vb);
            float u1 = sInfo.rnd(DIM_U1);
            float u2 = sInfo.rnd(DIM_U2);
            float phi = 2.0f * PI_F * u1;
            float cosTheta = rtx_powf(u2, 1.0f / (a + 1.0f));
            float sinTheta = std::sqrt(std::max(0.0f, 1.0f - cosTheta * cosTheta));
            V3 h = safe_normalize(U * (rtx_cosf(phi) * sinTheta) + Vb * (rtx_sinf(phi) * sinTheta) + ctx.N * cosTheta);
            if (dot(ctx.V, h) < 0.0f) h = -h;
            float cosI = -dot(h, I);
            float k = 1.0f - eta * eta * (1.0f - cosI * cosI);
            if (k < 0.0f) {
                dir = safe_normalize(reflect(-ctx.V, h));
                si.mult = transColor;
                si.prob = (pt / total);
                si.dist = -1.0f;
                si.lobe = LOBE_SPECULAR;
                return si.prob > 0.0f && vmax(si.mult) > 0.0f;
            }
            float cosT = std::sqrt(k);
            dir = safe_normalize(eta * I + (eta * cosI - cosT) * h);
            float absCosN = std::fabs(dot(ctx.N, dir));
            float VoH = dot(ctx.V, h);
            float DoH = dot(dir, h);
            if (absCosN <= 0.0f || VoH <= 1e-6f || VoH * DoH >= 0.0f) return false;
            float denom = eta * VoH + DoH;
            if (std::fabs(denom) <= 1e-6f) return false;
            float pdfH = ((a + 1.0f) / (2.0f * PI_F)) * rtx_powf(std::max(0.0f, dot(ctx.N, h)), a);
            float dwh_dwi = std::fabs((eta * eta * DoH) / (denom * denom));
            float pdfDir = pdfH * dwh_dwi;
            if (pdfDir <= 0.0f) return false;
            float etaScale = transmission_scale(eta);
            float NoH = std::max(0.0f, dot(ctx.N, h));
            V3 fTrans = transColor * ((a + 2.0f) / (2.0f * PI_F)) * rtx_powf(NoH, a);
            float factor = std::fabs(DoH * VoH) / std::max(1e-6f, std::fabs(dot(ctx.N, ctx.V)) * denom * denom);
            si.mult = fTrans * etaScale * factor;
            si.prob = (pt / total) * pdfDir;
            si.dist = 0.0f;
            si.lobe = LOBE_TRANSMISSION;
            return si.prob > 0.0f && vmax(si.mult) > 0.0f;
        }
    }
}

// Shading.cpp:533-713
static void evaluate_phong_blinn(const PhongCtx& ctx, bool phong, const SInfo& sInfo, V3 dirIn, DirInfo& si) {
    si.set_void();
    float etaI = sInfo.front ? 1.0f : ctx.ior;
    float etaT = sInfo.front ? ctx.ior : 1.0f;
    float F = (ctx.ior > 0.0f && vmax(ctx.kt) > 0.0f) ? fresnel_dielectric(std::fabs(dot(ctx.N, ctx.V)), etaI, etaT) : 0.0f;
    V3 fresnelRef = ctx.kt * F;
    V3 transColor = ctx.kt * (1.0f - F);
    V3 specColor = ctx.ks;
    if (vmax(ctx.kr) > 0.0f) specColor = V3(0.0f);
    specColor = clamp_max(specColor, 1.0f);

    V3 dir = safe_normalize(dirIn);
    float cosN = dot(ctx.N, dir);
    V3 mult(0.0f);
    float pdf = 0.0f;
    const float a = ctx.alpha;

    if (cosN > 0.0f) {
        float pr = luma1(ctx.kr + fresnelRef);
        float ps = luma1(specColor);
        float pd = luma1(ctx.kd);
        float pt = luma1(transColor);
        float total = pd + ps + pr + pt;
        if (total <= 0.0f) { si.set_void(); return; }
        float pdfDiff = pd > 0.0f ? (cosN / PI_F) * (pd / total) : 0.0f;
        float pdfRefl = 0.0f;
        V3 fRefl(0.0f);
        if (pr > 0.0f) {
            V3 reflColor = clamp_max(ctx.kr + fresnelRef, 1.0f);
            if (phong) {
                float cosR = std::max(0.0f, dot(ctx.R, dir));
                if (cosR > 0.0f) {
                    pdfRefl = ((a + 1.0f) / (2.0f * PI_F)) * rtx_powf(cosR, a);
                    pdfRefl *= pr / total;
                    fRefl = reflColor * ((a + 2.0f) / (2.0f * PI_F)) * rtx_powf(cosR, a);
                }
            } else {
                V3 H = safe_normalize(ctx.V + dir);
                float NoH = std::max(0.0f, dot(ctx.N, H));
                float VoH = std::max(0.0f, dot(ctx.V, H));
                if (NoH > 0.0f && VoH > 1e-6f) {
                    float pdfH = ((a + 1.0f) / (2.0f * PI_F)) * rtx_powf(NoH, a);
                    float pdfDir = pdfH / (4.0f * std::max(1e-6f, VoH));
                    pdfRefl = pdfDir * (pr / total);
                    float invVoH = 1.0f / std::max(1e-6f, VoH);
                    fRefl = reflColor * ((a + 8.0f) / (8.0f * PI_F)) * rtx_powf(NoH, a) * invVoH;
                }
            }
        }
        float pdfSpec = 0.0f;
        bool hasSpec = false;
        float blinnNoH = 0.0f, blinnVoH = 0.0f;
        if (!phong) {
            V3 H = safe_normalize(ctx.V + dir);
            blinnNoH = std::max(0.0f, dot(ctx.N, H));
            blinnVoH = std::max(0.0f, dot(ctx.V, H));
        }
        V3 fDiff = (pd > 0.0f) ? ctx.kd * (1.0f / PI_F) : V3(0.0f);
        V3 fSpec(0.0f);
        if (phong) {
            float cosR = std::max(0.0f, dot(ctx.R, dir));
            if (ps > 0.0f && cosR > 0.0f) {
                pdfSpec = ((a + 1.0f) / (2.0f * PI_F)) * rtx_powf(cosR, a);
                pdfSpec *= ps / total;
                hasSpec = true;
                fSpec = specColor * ((a + 2.0f) / (2.0f * PI_F)) * rtx_powf(cosR, a);
            }
        } else {
            if (ps > 0.0f && blinnNoH > 0.0f && blinnVoH > 1e-6f) {
                float pdfDir = ((a + 1.0f) / (2.0f * PI_F)) * rtx_powf(blinnNoH, a);
                pdfDir /= (4.0f * std::max(1e-6f, blinnVoH));
                pdfSpec = pdfDir * (ps / total);
                hasSpec = true;
                float invVoH = 1.0f / std::max(1e-6f, blinnVoH);
                fSpec = specColor * ((a + 8.0f) / (8.0f * PI_F)) * rtx_powf(blinnNoH, a) * invVoH;
            }
        }
        if (!phong) mult = (fRefl + fDiff + fSpec) * cosN;
        else mult = fRefl + (fDiff + fSpec) * cosN;
        pdf = pdfDiff + pdfSpec + pdfRefl;
        bool hasRefl = (pdfRefl > 0.0f);
        if ((hasSpec || hasRefl) && pd > 0.0f) si.lobe = LOBE_ALL;
        else if (hasSpec || hasRefl) si.lobe = LOBE_SPECULAR;
        else si.lobe = LOBE_DIFFUSE;
    } else {
        float pd = luma1(ctx.kd);
        float ps = luma1(specColor);
        float pr = (vmax(ctx.kr) > 0.0f) ? luma1(ctx.kr + fresnelRef) : 0.0f;
        float pt = luma1(transColor);
        float total = pd + ps + pr + pt;
        if (total <= 0.0f) { si.set_void(); return; }
        if (pt > 0.0f) {
            float eta = sInfo.front ? 1.0f / ctx.ior : ctx.ior;
            float absCosWo = std::fabs(dot(ctx.N, ctx.V));
            float absCosWi = std::fabs(dot(ctx.N, dir));
            if (absCosWi <= 0.0f || absCosWo <= 0.0f) { si.set_void(); return; }
            if (phong) {
                V3 I = -ctx.V;
                float cosI = -dot(ctx.N, I);
                float k = 1.0f - eta * eta * (1.0f - cosI * cosI);
                if (k < 0.0f) { si.set_void(); return; }
                float cosTIdeal = std::sqrt(k);
                V3 T = safe_normalize(eta * I + (eta * cosI - cosTIdeal) * ctx.N);
                float cosT = std::max(0.0f, dot(T, dir));
                if (cosT <= 0.0f) { si.set_void(); return; }
                float pdfDir = ((a + 1.0f) / (2.0f * PI_F)) * rtx_powf(cosT, a);
                float etaScale = transmission_scale(eta);
                V3 fTrans = transColor * ((a + 2.0f) / (2.0f * PI_F)) * rtx_powf(cosT, a);
                mult = fTrans * etaScale;
                pdf = (pt / total) * pdfDir;
                si.lobe = LOBE_TRANSMISSION;
            } else {
                V3 H = safe_normalize(ctx.V + dir * eta);
                if (dot(ctx.N, H) < 0.0f) H = -H;
                float NoH = std::max(0.0f, dot(ctx.N, H));
                if (NoH <= 0.0f) { si.set_void(); return; }
                float VoH = dot(ctx.V, H);
                float WiH = dot(dir, H);
                if (VoH * WiH >= 0.0f) { si.set_void(); return; }
                float denom = eta * VoH + WiH;
                if (std::fabs(denom) <= 1e-6f) { si.set_void(); return; }
                float pdfH = ((a + 1.0f) / (2.0f * PI_F)) * rtx_powf(NoH, a);
                float dwh_dwi = std::fabs((eta * eta * WiH) / (denom * denom));
                float pdfDir = pdfH * dwh_dwi;
                float etaScale = transmission_scale(eta);
                V3 fTrans = transColor * ((a + 2.0f) / (2.0f * PI_F)) * rtx_powf(NoH, a);
                float factor = std::fabs(WiH * VoH) / std::max(1e-6f, absCosWo * denom * denom);
                mult = fTrans * etaScale * factor;
                pdf = (pt / total) * pdfDir;
                si.lobe = LOBE_TRANSMISSION;
            }
        }
    }
    if (pdf <= 0.0f || vmax(mult) <= 0.0f) { si.set_void(); return; }
    si.mult = mult;
    si.prob = pdf;
    si.dist = 0.0f;
}

// ---- MtlMicrofacet (Shading.cpp:741-936) ----
static inline float ggx_d(float alpha2, float NoH) {   // :741-745
    float denom = NoH * NoH * (alpha2 - 1.0f) + 1.0f;
    return alpha2 / (PI_F * denom * denom);
}
static inline float ggx_g1(float alpha2, float NoX) {   // :747-751
    float denom = NoX + std::sqrt(alpha2 + (1.0f - alpha2) * NoX * NoX);
    return (2.0f * NoX) / denom;
}
static inline V3 fresnel_schlick(V3 F0, float VoH) {   // :758-762
    float k = rtx_powf(1.0f - VoH, 5.0f);
    return F0 + (V3(1.0f) - F0) * k;
}
struct MicroCtx {
    V3 N, V, F0, diffuse, trans;
    float alpha, ior, pd, ps, pt, total;
};
// common prologue of GenerateSample / GetSampleInfo (:768-785, :861-879); float maps scale by the texel's red channel
static MicroCtx build_micro(const Scene& sc, const rtx_material_desc& m, const SInfo& s) {
    MicroCtx c;
    c.N = normalized(s.N);
    if (!s.front) c.N = -c.N;
    c.V = safe_normalize(s.V);
    const V3 base = sc.eval_tc(m.diffuse, m.diffuse_tex, s.uvw);
    float rough = m.specular[0], metal = m.reflection[0];
    if (m.specular_tex >= 0) rough *= sc.texmaps[m.specular_tex].eval(s.uvw).x;
    if (m.reflection_tex >= 0) metal *= sc.texmaps[m.reflection_tex].eval(s.uvw).x;
    rough = clamp01(rough);
    metal = clamp01(metal);
    c.trans = sc.eval_tc(m.refraction, m.refraction_tex, s.uvw);
    c.alpha = std::max(0.002f, rough * rough);
    c.F0 = V3(0.04f) * (1.0f - metal) + base * metal;
    c.diffuse = base * (1.0f - metal);
    c.ior = m.ior;
    c.pd = luma1(c.diffuse);
    c.ps = luma1(c.F0);
    c.pt = luma1(c.trans);
    c.total = c.pd + c.ps + c.pt;
    return c;
}
static bool sample_microfacet(const MicroCtx& c, const SInfo& sInfo, V3& dir, DirInfo& si) {   // :765-856
    si.set_void();
    const V3 N = c.N, V = c.V;
    if (c.total <= 0.0f) return false;
    const float r = sInfo.rnd(DIM_LOBE) * c.total;
    const float u1 = sInfo.rnd(DIM_U1), u2 = sInfo.rnd(DIM_U2);
    if (r < c.pd) {
        V3 U, Vb;
        build_onb(N, U, Vb);
        float rxy = std::sqrt(u1);
        float phi = 2.0f * PI_F * u2;
        float x = rxy * rtx_cosf(phi);
        float y = rxy * rtx_sinf(phi);
        float z = std::sqrt(std::max(0.0f, 1.0f - u1));
        dir = safe_normalize(U * x + Vb * y + N * z);
        float cosN = std::max(0.0f, dot(N, dir));
        float pdf = cosN / PI_F;
        si.mult = c.diffuse * (1.0f / PI_F) * cosN;
        si.prob = (c.pd / c.total) * pdf;
        si.dist = 0.0f;
        si.lobe = LOBE_DIFFUSE;
        return si.prob > 0.0f && vmax(si.mult) > 0.0f;
    } else if (r < c.pd + c.ps) {
        V3 U, Vb;
        build_onb(N, U, Vb);
        float phi = 2.0f * PI_F * u1;
        float cosT = std::sqrt((1.0f - u2) / (1.0f + (c.alpha * c.alpha - 1.0f) * u2));
        float sinT = std::sqrt(std::max(0.0f, 1.0f - cosT * cosT));
        V3 h = safe_normalize(U * (rtx_cosf(phi) * sinT) + Vb * (rtx_sinf(phi) * sinT) + N * cosT);
        dir = reflect(-V, h);
        if (dot(N, dir) <= 0.0f) return false;
        float NoV = std::max(0.0f, dot(N, V));
        float NoL = std::max(0.0f, dot(N, dir));
        float NoH = std::max(0.0f, dot(N, h));
        float VoH = std::max(0.0f, dot(V, h));
        float alpha2 = c.alpha * c.alpha;
        float D = ggx_d(alpha2, NoH);
        float G = ggx_g1(alpha2, NoL) * ggx_g1(alpha2, NoV);
        V3 F = fresnel_schlick(c.F0, VoH);
        si.mult = (F * (D * G)) / std::max(1e-6f, 4.0f * NoV);
        float pdfDir = (D * NoH) / std::max(1e-6f, 4.0f * VoH);
        si.prob = (c.ps / c.total) * pdfDir;
        si.dist = 0.0f;
        si.lobe = LOBE_SPECULAR;
        return si.prob > 0.0f && vmax(si.mult) > 0.0f;
    } else {
        float eta = sInfo.front ? 1.0f / c.ior : c.ior;
        V3 I = -V;
        float cosI = clamp01(-dot(N, I));
        float k = 1.0f - eta * eta * (1.0f - cosI * cosI);
        if (k < 0.0f) return false;
        V3 T = eta * I + (eta * cosI - std::sqrt(k)) * N;
        dir = safe_normalize(T);
        float etaScale = transmission_scale(eta);
        si.mult = c.trans * etaScale;
        si.prob = c.pt / c.total;
        si.dist = -1.0f;
        si.lobe = LOBE_TRANSMISSION;
        return si.prob > 0.0f && vmax(si.mult) > 0.0f;
    }
}
static void evaluate_microfacet(const MicroCtx& c, const SInfo& sInfo, V3 dirIn, DirInfo& si) {   // :858-934
    si.set_void();
    const V3 N = c.N, V = c.V;
    const V3 L = safe_normalize(dirIn);
    if (c.total <= 0.0f) return;
    float NoL = dot(N, L);
    V3 mult(0.0f);
    float pdf = 0.0f;
    int lobe = LOBE_NONE;
    if (NoL > 0.0f) {
        V3 fDiff = c.diffuse * (1.0f / PI_F) * NoL;
        float pdfDiff = (c.pd > 0.0f) ? (NoL / PI_F) * (c.pd / c.total) : 0.0f;
        V3 H = safe_normalize(L + V);
        float NoV = std::max(0.0f, dot(N, V));
        float NoH = std::max(0.0f, dot(N, H));
        float VoH = std::max(0.0f, dot(V, H));
        float pdfSpec = 0.0f;
        V3 fSpec(0.0f);
        if (c.ps > 0.0f && VoH > 0.0f) {
            float alpha2 = c.alpha * c.alpha;
            float D = ggx_d(alpha2, NoH);
            float G = ggx_g1(alpha2, NoL) * ggx_g1(alpha2, NoV);
            V3 F = fresnel_schlick(c.F0, VoH);
            fSpec = (F * (D * G)) / std::max(1e-6f, 4.0f * NoV);
            pdfSpec = (D * NoH) / std::max(1e-6f, 4.0f * VoH);
            pdfSpec *= (c.ps / c.total);
        }
        mult = fDiff + fSpec;
        pdf = pdfDiff + pdfSpec;
        lobe = (c.ps > 0.0f && c.pd > 0.0f) ? LOBE_ALL : (c.ps > 0.0f ? LOBE_SPECULAR : LOBE_DIFFUSE);
    } else if (c.pt > 0.0f) {
        float eta = sInfo.front ? 1.0f / c.ior : c.ior;
        V3 I = -V;
        float cosI = -dot(N, I);
        float k = 1.0f - eta * eta * (1.0f - cosI * cosI);
        if (k < 0.0f) return;
        float cosT = std::sqrt(k);
        V3 T = safe_normalize(eta * I + (eta * cosI - cosT) * N);
        if (dot(T, L) < 1.0f - 1e-4f) return;
        mult = c.trans * transmission_scale(eta);
        pdf = c.pt / c.total;
        lobe = LOBE_TRANSMISSION;
    }
    if (pdf <= 0.0f || vmax(mult) <= 0.0f) return;
    si.mult = mult;
    si.prob = pdf;
    si.dist = 0.0f;
    si.lobe = lobe;
}

bool mtl_generate_sample(const Scene& sc, const Material& m, const SInfo& s, V3& dir, DirInfo& si) {
    if (m.d.type == RTX_MTL_MICROFACET) return sample_microfacet(build_micro(sc, m.d, s), s, dir, si);
    PhongCtx ctx = build_ctx(sc, m.d, s);  // Shading.cpp:715-719 / 727-731
    return sample_phong_blinn(ctx, m.d.type == RTX_MTL_PHONG, s, dir, si);
}
void mtl_get_sample_info(const Scene& sc, const Material& m, const SInfo& s, V3 dir, DirInfo& si) {
    if (m.d.type == RTX_MTL_MICROFACET) {
        evaluate_microfacet(build_micro(sc, m.d, s), s, dir, si);
        return;
    }
    PhongCtx ctx = build_ctx(sc, m.d, s);  // Shading.cpp:721-725 / 733-737
    evaluate_phong_blinn(ctx, m.d.type == RTX_MTL_PHONG, s, dir, si);
}
// GetDiffuseAlbedo / GetEmission / IsPhotonSurface as the path loop sees them (RayCaster.cpp:135-138, :158-161)
V3 mtl_diffuse_albedo(const Scene& sc, const Material& m, const SInfo& s) {
    if (m.d.type == RTX_MTL_MICROFACET) return V3(0.5f);
    return sc.eval_tc(m.d.diffuse, m.d.diffuse_tex, s.uvw);
}
V3 mtl_emission(const Scene& sc, const Material& m, const SInfo& s) {
    if (m.d.type == RTX_MTL_MICROFACET) return V3(0.0f);
    return sc.eval_tc(m.d.emission, m.d.emission_tex, s.uvw);
}
bool mtl_is_photon_surface(const Material& m) {
    const float k = m.d.type == RTX_MTL_MICROFACET ? 1.0f - clamp01(m.d.reflection[0]) : 1.0f;
    return (m.d.diffuse[0] + m.d.diffuse[1] + m.d.diffuse[2]) * k > 0.0f;
}

// BumpMapping.h:9-70
static inline void basis_from_normal(V3 n, V3& t, V3& b) {
    if (std::fabs(n.x) > 0.5f) t = cross(V3(0.0f, 1.0f, 0.0f), n);
    else t = cross(V3(1.0f, 0.0f, 0.0f), n);
    t = normalized(t);
    b = cross(n, t);
    b = normalized(b);
}
void apply_bump(const Scene& sc, const Material* mtl, Hit& h) {
    if (!mtl) return;
    if (mtl->d.bump_tex < 0) return;
    const TexMap& bump = sc.texmaps[mtl->d.bump_tex];
    V3 N = h.N;
    if (len2(N) <= 0.0f) return;
    N = normalized(N);
    V3 T = h.dpdu;
    V3 B = h.dpdv;
    if (len2(T) < 1e-10f || len2(B) < 1e-10f) {
        basis_from_normal(N, T, B);
    } else {
        T = T - N * dot(T, N);
        if (len2(T) < 1e-10f) basis_from_normal(N, T, B);
        else {
            T = normalized(T);
            B = B - N * dot(B, N);
            B = B - T * dot(B, T);
            if (len2(B) < 1e-10f) B = cross(N, T);
            B = normalized(B);
        }
    }
    float du = 0.001f, dv = 0.001f;
    int maxDim = std::max(bump.w, bump.h);
    if (maxDim > 0) du = dv = 1.0f / float(maxDim);
    const float h0 = luma1(bump.eval(h.uvw));
    const float hx = luma1(bump.eval(h.uvw + V3(du, 0.0f, 0.0f)));
    const float hy = luma1(bump.eval(h.uvw + V3(0.0f, dv, 0.0f)));
    const float invDu = (std::fabs(du) > 1e-6f) ? 1.0f / du : 0.0f;
    const float invDv = (std::fabs(dv) > 1e-6f) ? 1.0f / dv : 0.0f;
    const float bumpScale = mtl->d.bump_scale;
    const float dHdU = (hx - h0) * invDu * bumpScale;
    const float dHdV = (hy - h0) * invDv * bumpScale;
    V3 nTangent(-dHdU, -dHdV, 1.0f);
    if (len2(nTangent) > 0.0f) nTangent = normalized(nTangent);
    V3 bumped = T * nTangent.x + B * nTangent.y + N * nTangent.z;
    if (len2(bumped) > 0.0f) h.N = normalized(bumped);
}

// ---- visibility -----------------------------------------------------------------------------------------
static inline V3 offset_point(V3 p, V3 n, V3 dir) {  // RayCaster.cpp:108-112
    V3 nn = safe_normalize(n);
    float sign = dot(dir, nn) >= 0.0f ? 1.0f : -1.0f;
    return p + nn * (1e-4f * sign);
}

// RayCaster.cpp:178-278
float visible(const Scene& scene, V3 origin, V3 normal, V3 dir, float maxDist, bool useTransmittance, uint64_t* hops) {
    Ray shadowRay{offset_point(origin, normal, dir), safe_normalize(dir)};
    if (!useTransmittance) {  // VisibleSimple :178-190
        Hit h;
        h.init();
        h.z = maxDist;
        if (hops) (*hops)++;
        if (!scene.intersect_with_lights(shadowRay, h, RTX_HIT_FRONT_AND_BACK)) return 1.0f;
        if (h.light) return 1.0f;
        float dist = len(h.p - shadowRay.p);
        if (dist >= maxDist - 1e-4f) return 1.0f;
        return 0.0f;
    }
    const int kMaxHops = 8;
    float visibility = 1.0f;
    for (int hop = 0; hop < kMaxHops; ++hop) {
        Hit h;
        h.init();
        h.z = maxDist;
        if (hops) (*hops)++;
        if (!scene.intersect_with_lights(shadowRay, h, RTX_HIT_FRONT_AND_BACK)) return visibility;
        if (h.light) return visibility;
        float dist = len(h.p - shadowRay.p);
        if (dist >= maxDist - 1e-4f) return visibility;
        float transFactor = 0.0f;
        float ior = 1.5f;
        if (h.obj >= 0) {
            int mi = scene.meshes[h.obj].material;
            if (mi >= 0) {
                const rtx_material_desc& m = scene.materials[mi].d;
                // pb->Refraction().GetValue(): the TexturedColor value, not the texture
                transFactor = std::max(m.refraction[0], std::max(m.refraction[1], m.refraction[2]));
                if (transFactor <= 0.0f) return 0.0f;
                ior = m.ior;
            } else
                return 0.0f;
        } else
            return 0.0f;
        float cosTheta = std::fabs(dot(h.GN, -shadowRay.dir));
        float etaI = h.front ? 1.0f : ior;
        float etaT = h.front ? ior : 1.0f;
        cosTheta = std::min(std::max(cosTheta, 0.0f), 1.0f);
        float r0 = (etaI - etaT) / (etaI + etaT);
        r0 = r0 * r0;
        float F = r0 + (1.0f - r0) * rtx_powf(1.0f - cosTheta, 5.0f);
        float transmit = std::max(0.0f, std::min(1.0f, (1.0f - F) * transFactor));
        visibility *= std::max(0.0f, std::min(1.0f, transmit));
        if (visibility <= 1e-4f) return 0.0f;
        shadowRay.p = offset_point(h.p, h.GN, shadowRay.dir);
        maxDist -= dist;
        if (maxDist <= 1e-4f) return visibility;
    }
    return 0.0f;
}

// RayCaster.cpp:281-301
static inline V3 apply_absorption(V3 c, V3 absorption, float dist) {
    if (dist <= 0.0f) return c;
    if (absorption.x == 0.0f && absorption.y == 0.0f && absorption.z == 0.0f) return c;
    float absR = std::max(0.0f, absorption.x);
    float absG = std::max(0.0f, absorption.y);
    float absB = std::max(0.0f, absorption.z);
    const float ar = rtx_expf(-absR * dist);
    const float ag = rtx_expf(-absG * dist);
    const float ab = rtx_expf(-absB * dist);
    return V3(std::min(c.x, c.x * ar), std::min(c.y, c.y * ag), std::min(c.z, c.z * ab));
}

// RayCaster.cpp:303-330
static inline void dir_to_screen_uv(const Camera& cam, V3 dir, float& u, float& v) {
    V3 f = cam.dir;
    V3 r = normalized(cross(f, cam.up));
    V3 uvec = normalized(cross(r, f));
    float dz = dot(dir, f);
    if (std::fabs(dz) < 1e-6f) dz = (dz >= 0.0f) ? 1e-6f : -1e-6f;
    float sx = dot(dir, r) / dz;
    float sy = dot(dir, uvec) / dz;
    float fovRad = cam.fov * PI_F / 180.0f;
    float halfY = rtx_tanf(0.5f * fovRad);
    float aspect = (cam.H > 0) ? float(cam.W) / float(cam.H) : 1.0f;
    float halfX = halfY * aspect;
    float ndcX = sx / halfX;
    float ndcY = sy / halfY;
    u = (ndcX + 1.0f) * 0.5f;
    v = (1.0f - ndcY) * 0.5f;
    u = u - std::floor(u);
    v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
}

static inline float power_heuristic(float pdfA, float pdfB) {  // RayCaster.cpp:102-106
    float a = pdfA * pdfA;
    float b = pdfB * pdfB;
    return a / (a + b);
}

static inline bool has_map_or_value(V3 c, int tex) { return tex >= 0 || c.x != 0.0f || c.y != 0.0f || c.z != 0.0f; }

// the miss branch of TracePath (RayCaster.cpp:362-382) and of TraceLegacy (:583-593, camera rays only)
V3 miss_radiance(const Scene& scene, const Camera& camera, V3 nd, bool cameraRay) {
    float uEnv = 0.0f, vEnv = 0.0f;
    dir_to_screen_uv(camera, nd, uEnv, vEnv);
    const bool hasBackground = has_map_or_value(scene.bgColor, scene.bgTex);
    const bool hasEnvironment = has_map_or_value(scene.envColor, scene.envTex);
    float bgv[3] = {scene.bgColor.x, scene.bgColor.y, scene.bgColor.z};
    const V3 backplate = hasBackground ? scene.eval_tc(bgv, scene.bgTex, V3(uEnv, vEnv, 0.0f)) : V3(0, 0, 0);
    V3 envLight(0, 0, 0);
    if (hasEnvironment) envLight = scene.envTex >= 0 ? scene.envColor * scene.texmaps[scene.envTex].eval_env(nd) : scene.envColor;
    const V3 bgEmit = (hasBackground && scene.bgEmission > 0.0f) ? backplate * scene.bgEmission : V3(0, 0, 0);
    if (cameraRay) {
        const V3 vis = hasBackground ? backplate : envLight;
        return vis + bgEmit;
    }
    return envLight + bgEmit;
}

// RayCaster.cpp:339-574
PathResult trace_path(const Scene& scene, const Camera& camera, Ray ray, int px, int py, int sampleIdx,
                      const PhotonMapper* mapper, int maxBounces, uint64_t seed, PathCounters* pc) {
    const float kMinPdf = 1e-4f;
    const float kRadianceClamp = 10.0f;
    const float kCausticBoost = 0.1f;
    V3 L(0, 0, 0);
    V3 beta(1, 1, 1);
    float primaryDist = BIGFLOAT;
    V3 firstAlbedo(0, 0, 0), firstNormal(0, 0, 0);
    bool haveFirstHit = false, firstHit = true, lastDelta = true;
    float lastPdf = 1.0f;
    SInfo sInfo, prevSInfo;
    bool hasPrevSInfo = false;
    bool insideMedium = false;
    V3 mediumAbsorption(0.0f);
    const uint32_t pixelIdx = (uint32_t)(py * camera.W + px);

    for (int bounce = 0; bounce < maxBounces; ++bounce) {
        Hit h;
        h.init();
        if (pc) (bounce == 0 ? pc->primary : pc->bounce)++;
        bool hitGeom = scene.intersect_with_lights(ray, h, RTX_HIT_FRONT_AND_BACK);
        if (!hitGeom) {
            L = L + beta * miss_radiance(scene, camera, normalized(ray.dir), bounce == 0);
            break;
        }
        float hitDist = len(h.p - ray.p);
        if (insideMedium && hitDist > 0.0f) beta = apply_absorption(beta, mediumAbsorption, hitDist);

        sInfo.seed = seed;
        sInfo.ka = pixelIdx;
        sInfo.kb = (uint32_t)sampleIdx;
        sInfo.stream = (uint32_t)bounce;
        const Material* mtl = (h.obj >= 0 && scene.meshes[h.obj].material >= 0) ? &scene.materials[scene.meshes[h.obj].material] : nullptr;
        apply_bump(scene, mtl, h);
        sInfo.set_hit(ray, h);

        if (firstHit) {
            primaryDist = hitDist;
            firstHit = false;
            firstAlbedo = mtl ? mtl_diffuse_albedo(scene, *mtl, sInfo) : V3(0.5f);  // :135-138
            firstNormal = safe_normalize(h.GN);
            haveFirstHit = true;
        }

        if (h.light && h.lightIdx >= 0) {
            const Light& light = scene.lights[h.lightIdx];
            V3 emitted = light_radiance(light);
            float lightPdfSA = 0.0f;
            if (hasPrevSInfo) {
                DirInfo li;
                light_get_sample_info(light, prevSInfo.P, ray.dir, li);
                lightPdfSA = li.prob;
                if (light.renderable()) {
                    float cosLight = std::max(0.0f, dot(-h.N, safe_normalize(ray.dir)));
                    float dist2 = li.dist * li.dist;
                    if (cosLight > 1e-6f && dist2 > 0.0f) lightPdfSA = li.prob * dist2 / cosLight;
                    else lightPdfSA = 0.0f;
                } else
                    lightPdfSA = 0.0f;
            }
            float w = (lastDelta || lightPdfSA <= 0.0f) ? 1.0f : power_heuristic(lastPdf, lightPdfSA);
            L = L + beta * emitted * w;
            break;
        }
        if (!mtl) break;

        V3 emission = mtl_emission(scene, *mtl, sInfo);
        if (vmax(emission) > 0.0f) L = L + beta * emission;

        V3 direct(0, 0, 0);
        float maxVisibility = 0.0f;
        const bool useTransmittance = (bounce <= 8);
        for (size_t li_ = 0; li_ < scene.lights.size(); ++li_) {
            const Light& light = scene.lights[li_];
            V3 ldir;
            DirInfo li;
            if (!light_generate_sample(light, sInfo, (int)li_, ldir, li)) continue;
            if (li.prob <= 0.0f || vmax(li.mult) <= 0.0f) continue;
            float shadowMax = (li.dist > 0.0f && li.dist < BIGFLOAT * 0.5f) ? li.dist : BIGFLOAT;
            uint64_t hops = 0;
            float transmittance = visible(scene, h.p, sInfo.GN, ldir, shadowMax, useTransmittance, &hops);
            if (pc) pc->shadow += hops;
            maxVisibility = std::max(maxVisibility, transmittance);
            if (transmittance <= 0.0f) continue;
            DirInfo bsdf;
            mtl_get_sample_info(scene, *mtl, sInfo, safe_normalize(ldir), bsdf);
            if (bsdf.prob <= 0.0f || vmax(bsdf.mult) <= 0.0f) continue;
            float w = 1.0f;
            if (light.renderable()) {
                float dist = li.dist;
                float cosLight = 1.0f;
                Ray lightRay{h.p, safe_normalize(ldir)};
                Hit lightHit;
                lightHit.init();
                if (light.intersect(lightRay, lightHit)) cosLight = std::max(0.0f, dot(-lightHit.N, safe_normalize(ldir)));
                float pdfSolidAngle = (cosLight > 1e-6f && dist > 1e-6f) ? li.prob * dist * dist / cosLight : 0.0f;
                if (pdfSolidAngle <= 0.0f) continue;
                w = power_heuristic(pdfSolidAngle, bsdf.prob);
                float G = 1.0f / std::max(1e-6f, dist * dist);
                direct = direct + li.mult * bsdf.mult * G * (w / std::max(1e-6f, li.prob)) * transmittance;
            } else {
                w = power_heuristic(li.prob, bsdf.prob);
                direct = direct + li.mult * bsdf.mult * (w / std::max(kMinPdf, li.prob)) * transmittance;
            }
        }
        if (vmax(direct) > 0.0f) L = L + beta * direct;

        if (mapper) {
            const PhotonMap& cm = mapper->caustic;
            bool photonSurface = mtl_is_photon_surface(*mtl);  // IsPhotonSurface
            if (!cm.ph.empty() && photonSurface) {
                int maxPhotons = mapper->searchCount <= 30 ? 30 : (mapper->searchCount <= 50 ? 50 : 100);
                V3 irrad = cm.estimate(maxPhotons, mapper->causticRadius, h.p, h.N);
                if (vmax(irrad) > 0.0f) {
                    V3 kd = mtl_diffuse_albedo(scene, *mtl, sInfo);
                    irrad = irrad * kCausticBoost;
                    float causticScale = 0.1f + 0.9f * maxVisibility;
                    L = L + beta * (kd * (1.0f / PI_F) * irrad * causticScale);
                }
            }
        }

        V3 nextDir;
        DirInfo bs;
        if (!mtl_generate_sample(scene, *mtl, sInfo, nextDir, bs)) break;
        if (bs.prob <= 0.0f || vmax(bs.mult) <= 0.0f) break;
        float pdfThroughput = std::max(kMinPdf, bs.prob);
        beta = beta * (bs.mult / pdfThroughput);
        lastPdf = bs.prob;
        lastDelta = (bs.dist < 0.0f);
        prevSInfo = sInfo;
        hasPrevSInfo = true;
        if (bs.lobe & LOBE_TRANSMISSION) {
            if (sInfo.front) {
                insideMedium = true;
                mediumAbsorption = V3(mtl->d.absorption[0], mtl->d.absorption[1], mtl->d.absorption[2]);
            } else {
                insideMedium = false;
                mediumAbsorption = V3(0.0f);
            }
        }
        if (bounce >= 5) {
            float q = clamp01(vmax(beta));
            q = std::max(q, 0.05f);
            if (sInfo.rnd(DIM_RR) > q) break;
            beta = beta * (1.0f / q);
        }
        ray.p = offset_point(h.p, sInfo.GN, nextDir);
        ray.dir = safe_normalize(nextDir);
    }
    if (kRadianceClamp > 0.0f) L = clamp_max(L, kRadianceClamp);
    return PathResult{L, primaryDist, firstAlbedo, firstNormal, haveFirstHit};
}

// TraceRadiance (RayCaster.cpp:634-640): the compile-time SHADING_USE_LEGACY switch is rtx_render_params::shading_mode
PathResult trace_radiance(const Scene& sc, const Camera& cam, Ray ray, int px, int py, int sampleIdx, const PhotonMapper* mapper,
                          const rtx_render_params& rp, PathCounters* pc) {
    if (rp.shading_mode == RTX_SHADING_LEGACY) return trace_legacy(sc, cam, ray, px, py, sampleIdx, mapper, rp.seed, pc);
    return trace_path(sc, cam, ray, px, py, sampleIdx, mapper, rp.max_bounces, rp.seed, pc);
}

// ---- film / RenderBucket ---------------------------------------------------------------------------------
void Film::init(int w, int h) {
    W = w;
    H = h;
    accum.assign((size_t)w * h * 4, 0.0f);
    count.assign((size_t)w * h, 0);
    z.assign((size_t)w * h, BIGFLOAT);
    albedo.assign((size_t)w * h * 4, 0.0f);
    normal.assign((size_t)w * h * 4, 0.0f);
}

static inline bool ci_exceeds(int n, const float sum[3], const float sum2[3], float maxHalf) {  // :848-868 / :957-975
    float meanR = sum[0] / float(n), meanG = sum[1] / float(n), meanB = sum[2] / float(n);
    float varR = sum2[0] / float(n) - meanR * meanR;
    float varG = sum2[1] / float(n) - meanG * meanG;
    float varB = sum2[2] / float(n) - meanB * meanB;
    float vm = std::max(varR, std::max(varG, varB));
    if (vm < 0.0f) vm = 0.0f;
    float stddev = std::sqrt(vm);
    int df = n - 1;
    float t;
    if (df <= 3) t = 3.182f;
    else if (df <= 7) t = 2.365f;
    else if (df <= 15) t = 2.131f;
    else if (df <= 30) t = 2.042f;
    else t = 1.96f;
    float half_width = t * stddev / std::sqrt(float(n));
    return half_width > maxHalf;
}

struct PixelCtx {
    const Scene* sc;
    const Camera* cam;
    const rtx_render_params* rp;
    const PhotonMapper* mapper;
    Film* film;
    PathCounters* pc;
};

// one sample: RayCaster.cpp:783-838 (and the identical refinement body :899-952)
static void take_sample(const PixelCtx& c, int x, int y, int s, float shiftX, float shiftY, float& bestZ, int& nSamples,
                        float sum[3], float sum2[3]) {
    const int W = c.cam->W, H = c.cam->H;
    const int stride = c.rp->index_stride > 0 ? c.rp->index_stride : 1024;
    int sampleIndex = (y * W + x) * stride + (s + 1);
    float hx = halton(sampleIndex, 2);
    float hy = halton(sampleIndex, 3);
    float jitterX = std::fmod(hx + shiftX, 1.0f) - 0.5f;
    float jitterY = std::fmod(hy + shiftY, 1.0f) - 0.5f;
    float samplePosX = float(x) + 0.5f + jitterX;
    float samplePosY = float(y) + 0.5f + jitterY;
    Ray ray = generate_ray(*c.cam, samplePosX, samplePosY, sampleIndex);
    PathResult pr = trace_radiance(*c.sc, *c.cam, ray, x, y, s, c.mapper, *c.rp, c.pc);
    V3 col = pr.radiance;
    if (pr.primaryDist < bestZ) bestZ = pr.primaryDist;
    Film& f = *c.film;
    int origIdx = y * W + x;
    f.count[origIdx] += 1;
    if (pr.hit && f.albedo[origIdx * 4 + 3] == 0.0f) {  // StoreGuide :709-714
        f.albedo[origIdx * 4 + 3] = 1.0f;
        f.albedo[origIdx * 4 + 0] = pr.albedo.x;
        f.albedo[origIdx * 4 + 1] = pr.albedo.y;
        f.albedo[origIdx * 4 + 2] = pr.albedo.z;
        V3 n = safe_normalize(pr.normal);
        f.normal[origIdx * 4 + 0] = n.x;
        f.normal[origIdx * 4 + 1] = n.y;
        f.normal[origIdx * 4 + 2] = n.z;
        f.normal[origIdx * 4 + 3] = 1.0f;
    }
    float radius = filter_radius(c.rp->filter_type, c.rp->filter_radius);
    int x0 = std::max(0, int(std::floor(samplePosX - radius)));
    int x1 = std::min(W - 1, int(std::ceil(samplePosX + radius)));
    int y0 = std::max(0, int(std::floor(samplePosY - radius)));
    int y1 = std::min(H - 1, int(std::ceil(samplePosY + radius)));
    for (int yy = y0; yy <= y1; ++yy) {
        float wy = filter1d(c.rp->filter_type, c.rp->filter_radius, float(yy) + 0.5f - samplePosY);
        if (wy <= 0.0f) continue;
        for (int xx = x0; xx <= x1; ++xx) {
            float wx = filter1d(c.rp->filter_type, c.rp->filter_radius, float(xx) + 0.5f - samplePosX);
            if (wx <= 0.0f) continue;
            float w = wx * wy;
            size_t idx = (size_t)yy * W + xx;
            f.accum[idx * 4 + 0] += col.x * w;
            f.accum[idx * 4 + 1] += col.y * w;
            f.accum[idx * 4 + 2] += col.z * w;
            f.accum[idx * 4 + 3] += w;
            if (bestZ < f.z[idx]) f.z[idx] = bestZ;
        }
    }
    nSamples += 1;
    sum[0] += col.x;
    sum[1] += col.y;
    sum[2] += col.z;
    sum2[0] += col.x * col.x;
    sum2[1] += col.y * col.y;
    sum2[2] += col.z * col.z;
}

// per-pixel view of RenderBucket :774-980 (the outcome for a pixel depends only on its own samples)
static void render_pixel(const PixelCtx& c, int x, int y) {
    float shiftX, shiftY;
    pixel_shifts(x, y, shiftX, shiftY);
    int n = 0;
    float sum[3] = {0, 0, 0}, sum2[3] = {0, 0, 0};
    float bestZ = BIGFLOAT;
    const int initial = c.rp->spp_min;
    for (int s = 0; s < initial; ++s) take_sample(c, x, y, s, shiftX, shiftY, bestZ, n, sum, sum2);
    if (c.rp->spp_max <= c.rp->spp_min || n <= 1) return;
    if (!ci_exceeds(n, sum, sum2, c.rp->ci_halfwidth)) return;
    bestZ = BIGFLOAT;  // :893
    int samplesTaken = n;
    while (samplesTaken < c.rp->spp_max) {
        int toAdd = std::min(c.rp->spp_min, c.rp->spp_max - samplesTaken);
        for (int i = 0; i < toAdd; ++i) {
            take_sample(c, x, y, samplesTaken, shiftX, shiftY, bestZ, n, sum, sum2);
            ++samplesTaken;
        }
        if (toAdd <= 0) break;
        if (!ci_exceeds(n, sum, sum2, c.rp->ci_halfwidth)) break;
    }
}

// CastRays :1146-1314 — buckets row-major; bucket b is rendered when b % world == rank (tile sharding) and,
// for the bounded CPU-baseline subset, when its bucket coordinates are multiples of subset_stride.
void render(const Scene& sc, const Camera& cam, const rtx_render_params& rp, const PhotonMapper* mapper, Film& film,
            int nthreads, int subset_stride, PathCounters* pcOut) {
    const int W = cam.W, H = cam.H;
    const int bs = rp.tile_size > 0 ? rp.tile_size : 16;
    struct B {
        int x0, y0, x1, y1;
    };
    std::vector<B> buckets;
    int world = rp.world > 0 ? rp.world : 1;
    int bi = 0;
    for (int y = 0; y < H; y += bs)
        for (int x = 0; x < W; x += bs, ++bi) {
            if (bi % world != rp.rank) continue;
            if (subset_stride > 1 && (((x / bs) % subset_stride) != 0 || ((y / bs) % subset_stride) != 0)) continue;
            buckets.push_back(B{x, y, std::min(x + bs, W), std::min(y + bs, H)});
        }
    if (nthreads < 1) nthreads = 1;
    std::vector<Film> films(nthreads);
    std::vector<PathCounters> pcs(nthreads);
    std::vector<std::thread> workers;
    for (int t = 0; t < nthreads; ++t) {
        films[t].init(W, H);
        workers.emplace_back([&, t]() {
            PixelCtx c{&sc, &cam, &rp, mapper, &films[t], &pcs[t]};
            for (size_t b = t; b < buckets.size(); b += nthreads)
                for (int y = buckets[b].y0; y < buckets[b].y1; ++y)
                    for (int x = buckets[b].x0; x < buckets[b].x1; ++x) render_pixel(c, x, y);
        });
    }
    for (auto& th : workers) th.join();
    // merge in thread order (mergeToImage :1188-1200 sums per-thread buffers)
    for (int t = 0; t < nthreads; ++t) {
        const Film& f = films[t];
        for (size_t i = 0; i < (size_t)W * H; ++i) {
            for (int k = 0; k < 4; ++k) film.accum[i * 4 + k] += f.accum[i * 4 + k];
            film.count[i] += f.count[i];
            if (f.z[i] < film.z[i]) film.z[i] = f.z[i];
            if (film.albedo[i * 4 + 3] == 0.0f && f.albedo[i * 4 + 3] != 0.0f) {
                for (int k = 0; k < 4; ++k) {
                    film.albedo[i * 4 + k] = f.albedo[i * 4 + k];
                    film.normal[i * 4 + k] = f.normal[i * 4 + k];
                }
            }
        }
        if (pcOut) {
            pcOut->primary += pcs[t].primary;
            pcOut->bounce += pcs[t].bounce;
            pcOut->shadow += pcs[t].shadow;
        }
    }
}

// ---- photon map -----------------------------------------------------------------------------------------
void PhotonMap::scale(float s) {
    for (auto& p : ph)
        for (int k = 0; k < 3; ++k) p.power[k] *= s;
}

// Stand-in for cyPhotonMap's balanced kd-tree (absent): a uniform grid. The grid is a window of the lattice
// floor(p / cell) anchored at the origin; its edge is the search radius, doubled until the photons' bounding box spans
// at most 512 cells per axis and 2^24 in all. Photons are kept cell by cell and, inside a cell, in emission order
// ((photon, bounce) ascending). This fixes the order of the floating-point sum in estimate() below.
static int lattice_clamped(float x, float cell, int imin, int g) {
    return (int)std::fmin(std::fmax(std::floor(x / cell) - (float)imin, 0.0f), (float)(g - 1));
}

void PhotonMap::prepare(float radius) {
    cellStart.clear();
    order.clear();
    if (ph.empty()) return;
    Box b;
    for (auto& p : ph) b.add(V3(p.pos[0], p.pos[1], p.pos[2]));
    const float lo[3] = {b.pmin.x, b.pmin.y, b.pmin.z}, hi[3] = {b.pmax.x, b.pmax.y, b.pmax.z};
    cell = std::max(radius, 1e-6f);
    for (;; cell *= 2.0f) {
        double total = 1.0;
        bool ok = true;
        for (int a = 0; a < 3; ++a) {
            const float f0 = std::floor(lo[a] / cell), f1 = std::floor(hi[a] / cell);
            const double d = (double)f1 - (double)f0 + 1.0;
            if (!(d <= 512.0) || !(std::fabs(f0) < 4194304.0f) || !(std::fabs(f1) < 4194304.0f)) ok = false;
            total *= d;
        }
        if (ok && total <= 16777216.0) break;
        if (!(cell < 1e30f)) {   // non-finite positions: one cell
            break;
        }
    }
    int g[3];
    for (int a = 0; a < 3; ++a) {
        imin[a] = (int)std::floor(lo[a] / cell);
        g[a] = std::max(1, (int)std::floor(hi[a] / cell) - imin[a] + 1);
    }
    gx = g[0];
    gy = g[1];
    gz = g[2];
    std::vector<int> cnt((size_t)gx * gy * gz + 1, 0);
    auto cellOf = [&](const rtx_photon& p) {
        const int cx = lattice_clamped(p.pos[0], cell, imin[0], gx);
        const int cy = lattice_clamped(p.pos[1], cell, imin[1], gy);
        const int cz = lattice_clamped(p.pos[2], cell, imin[2], gz);
        return ((size_t)cz * gy + cy) * gx + cx;
    };
    for (auto& p : ph) cnt[cellOf(p) + 1]++;
    for (size_t i = 1; i < cnt.size(); ++i) cnt[i] += cnt[i - 1];
    cellStart = cnt;
    order.resize(ph.size());
    std::vector<int> fill(cnt.begin(), cnt.end() - 1);
    for (size_t i = 0; i < ph.size(); ++i) order[fill[cellOf(ph[i])]++] = (int)i;   // ascending index inside a cell
}

// PhotonMap::EstimateIrradiance<maxPhotons, PHOTONMAP_FILTER_LINEAR> (cyPhotonMap, absent). Stand-in
// definition: the k nearest photons within `radius` (ties by index); r = distance of the k-th when
// k were found, else `radius`; photons arriving against the normal (dir.N < 0) weighted by the cone filter
// 1 - d/r; normalisation 1 / ((1 - 2/3) pi r^2). The sum runs over the selected photons in the order of the grid
// (cells z-major, then y, then x; inside a cell by index).
V3 PhotonMap::estimate(int maxPhotons, float radius, V3 pos, V3 normal) const {
    if (ph.empty() || order.empty()) return V3(0, 0, 0);
    struct Cand {
        float d2;
        int idx;
    };
    std::vector<Cand> cand;
    float r2 = radius * radius;
    const int cx0 = lattice_clamped(pos.x - radius, cell, imin[0], gx), cx1 = lattice_clamped(pos.x + radius, cell, imin[0], gx);
    const int cy0 = lattice_clamped(pos.y - radius, cell, imin[1], gy), cy1 = lattice_clamped(pos.y + radius, cell, imin[1], gy);
    const int cz0 = lattice_clamped(pos.z - radius, cell, imin[2], gz), cz1 = lattice_clamped(pos.z + radius, cell, imin[2], gz);
    for (int cz = cz0; cz <= cz1; ++cz)
        for (int cy = cy0; cy <= cy1; ++cy)
            for (int cx = cx0; cx <= cx1; ++cx) {
                size_t c = ((size_t)cz * gy + cy) * gx + cx;
                for (int k = cellStart[c]; k < cellStart[c + 1]; ++k) {
                    const rtx_photon& p = ph[order[k]];
                    V3 d = V3(p.pos[0], p.pos[1], p.pos[2]) - pos;
                    float d2 = len2(d);
                    if (d2 <= r2) cand.push_back(Cand{d2, order[k]});
                }
            }
    if (cand.empty()) return V3(0, 0, 0);
    auto less = [](const Cand& a, const Cand& b) { return a.d2 < b.d2 || (a.d2 == b.d2 && a.idx < b.idx); };
    float rr2 = r2;
    Cand last{r2, 0x7FFFFFFF};   // the k-th nearest: everything up to it is selected
    if ((int)cand.size() >= maxPhotons) {
        std::vector<Cand> sel(cand);
        std::nth_element(sel.begin(), sel.begin() + (maxPhotons - 1), sel.end(), less);
        last = sel[maxPhotons - 1];
        rr2 = last.d2;
    }
    if (rr2 <= 0.0f) return V3(0, 0, 0);
    float r = std::sqrt(rr2);
    V3 irrad(0, 0, 0);
    for (auto& c : cand) {   // grid order
        if (less(last, c)) continue;
        const rtx_photon& p = ph[c.idx];
        if (dot(V3(p.dir[0], p.dir[1], p.dir[2]), normal) >= 0.0f) continue;
        float w = 1.0f - std::sqrt(c.d2) / r;
        irrad = irrad + V3(p.power[0], p.power[1], p.power[2]) * w;
    }
    return irrad * (3.0f / (PI_F * rr2));
}

// PhotonMapBuilder.cpp:149-268, recursion unrolled into a loop (tail call)
static void trace_photon(const Scene& scene, PhotonMapper& pm, uint64_t seed, uint32_t photonId, Ray ray, V3 power,
                         int maxBounces) {
    bool causticPath = false, hasHitDiffuse = false, hasTransmitted = false;
    for (int bounce = 0;; ++bounce) {
        if (bounce > maxBounces) return;
        if (vmax(power) < 1e-6f) return;
        Hit hInfo;
        hInfo.init();
        pm.rays++;
        if (!scene.intersect_node(ray, hInfo, RTX_HIT_FRONT_AND_BACK, true)) return;
        int mi = scene.meshes[hInfo.obj].material;
        if (mi < 0) return;
        const Material* mtl = &scene.materials[mi];
        apply_bump(scene, mtl, hInfo);
        V3 hitPos = hInfo.p;
        V3 N = normalized(hInfo.N);
        bool inside = dot(ray.dir, N) > 0.0f;
        bool isDiffuseSurface = mtl_is_photon_surface(*mtl);
        bool storeAsCaustic = causticPath && hasTransmitted && !hasHitDiffuse && isDiffuseSurface;
        if (bounce >= 1 && isDiffuseSurface) {
            V3 d = normalized(ray.dir);
            rtx_photon rec{{hitPos.x, hitPos.y, hitPos.z}, {d.x, d.y, d.z}, {power.x, power.y, power.z}};
            if (storeAsCaustic) pm.caustic.ph.push_back(rec);
            else pm.global.ph.push_back(rec);
            hasHitDiffuse = true;
            causticPath = false;
        }
        SInfo s;
        s.P = hInfo.p;
        s.V = -normalized(ray.dir);
        s.N = hInfo.N;
        s.GN = hInfo.GN;
        s.uvw = hInfo.uvw;
        s.front = !inside;
        s.mtlID = hInfo.mtlID;
        s.seed = seed;
        s.ka = photonId;
        s.kb = 0xFFFFFFFFu;
        s.stream = (uint32_t)(bounce + 1);
        V3 nextDir;
        DirInfo si;
        if (!mtl_generate_sample(scene, *mtl, s, nextDir, si)) return;
        if (!hasHitDiffuse && si.lobe == LOBE_SPECULAR && vmax(si.mult) > 0.0f) return;
        V3 adjustedPower = power;
        {
            float continueProbability = std::min(0.5f, vmax(si.mult));
            if (continueProbability < 1e-6f) return;
            if (s.rnd(DIM_RR) > continueProbability) return;
            adjustedPower = power * (1.0f / continueProbability);
        }
        V3 nextPower = adjustedPower * si.mult;
        if (si.prob > 1e-6f) nextPower = nextPower * (1.0f / si.prob);
        else return;
        bool nextCaustic = causticPath;
        bool nextHasTransmitted = hasTransmitted;
        if (si.lobe == LOBE_DIFFUSE) nextCaustic = false;
        else if (si.lobe == LOBE_TRANSMISSION && !hasHitDiffuse) {
            nextCaustic = true;
            nextHasTransmitted = true;
        }
        V3 Nn = inside ? -N : N;
        V3 biasDir = normalized(nextDir);
        float biasAmount = 1e-4f;
        V3 nextOrigin = hitPos + (dot(Nn, biasDir) >= 0.0f ? Nn : -Nn) * biasAmount;
        ray = Ray{nextOrigin, biasDir};
        power = nextPower;
        causticPath = nextCaustic;
        hasTransmitted = nextHasTransmitted;
    }
}

// PhotonMapBuilder.cpp:20-147. Photons are emitted in id order by one thread with the counter RNG keyed by
// photon id (the reference hands ids to threads dynamically and uses per-thread RNG streams).
PhotonMapper* photon_build(const Scene& scene, const rtx_photon_params* pp) {
    PhotonMapper* pm = new PhotonMapper();
    pm->searchCount = pp->search_count > 0 ? pp->search_count : 100;
    // AdjustPhotonGatherSettings :62-78
    if (!scene.rootBox.empty()) {
        V3 extent = scene.rootBox.pmax - scene.rootBox.pmin;
        float maxExtent = std::max(std::fabs(extent.x), std::max(std::fabs(extent.y), std::fabs(extent.z)));
        if (maxExtent > 0.0f) {
            if (pp->photon_radius <= 0.0f) pm->photonRadius = std::max(0.05f, maxExtent * 0.02f);
            else pm->photonRadius = pp->photon_radius;
            if (pp->caustic_radius <= 0.0f) pm->causticRadius = std::max(0.01f, pm->photonRadius * 0.25f);
            else pm->causticRadius = pp->caustic_radius;
        }
    }
    float totalIntensity = 0.0f;
    std::vector<float> lightIntensities;
    for (auto& L : scene.lights) {
        float power = L.intensity.x + L.intensity.y + L.intensity.z;
        lightIntensities.push_back(power);
        totalIntensity += power;
    }
    if (totalIntensity > 0.0f) {
        int stride = pp->photon_stride > 0 ? pp->photon_stride : 1;
        for (int photonId = pp->first_photon; photonId < pp->num_photons; photonId += stride) {
            float r = rnd01(pp->seed, (uint32_t)photonId, 0xFFFFFFFFu, 0, 0) * totalIntensity;
            float sum = 0.0f;
            int lightIndex = 0;
            for (size_t j = 0; j < scene.lights.size(); ++j) {
                sum += lightIntensities[j];
                if (r <= sum) {
                    lightIndex = (int)j;
                    break;
                }
            }
            const Light& light = scene.lights[lightIndex];
            float u[4];
            for (int k = 0; k < 4; ++k) u[k] = rnd01(pp->seed, (uint32_t)photonId, 0xFFFFFFFFu, 0, 1 + k);
            Ray photonRay;
            V3 photonPower;
            light_random_photon(light, u, photonRay, photonPower);
            float lightSelectionPdf = lightIntensities[lightIndex] / totalIntensity;
            V3 scaledPower = photonPower * (1.0f / lightSelectionPdf);
            trace_photon(scene, *pm, pp->seed, (uint32_t)photonId, photonRay, scaledPower, pp->max_bounces);
        }
    }
    return pm;
}

}  // namespace orc
