// Mirror of the ray queries of the reference's RayTracer/ShadeInfoAdapter.h (legacy shading mode,
// SHADING_USE_LEGACY=1): TraceShadowRay with its bias rules (:49-92), evaluated with one closest-hit on the GPU, and
// TraceSecondaryRay (:94-231), which recurses into the legacy Shade() functions of Shading_Legacy.cpp on the GPU
// (rtx_trace_secondary). Whole frames in this mode go through RayCaster with sampling.legacyShading = true.
#pragma once
#include <cmath>
#include <iostream>
#include "IntersectNodeAdapter.h"
#include "PhotonMapBuilder.h"

namespace rtamd {

// Stand-ins for the framework types in the reference's constructor signature (rng.h, scene.h: ShadeInfo are not in the
// reference repository). The random numbers of the GPU path come from the counter RNG keyed by (seed, pixel, sample,
// recursion node) in both the oracle and the kernels (deviation 1), so the RNG& is accepted and not drawn from.
class RNG {
public:
    RNG() = default;
    explicit RNG(uint64_t s) : state(s) {}
    void SetSequence(uint64_t s) { state = s; }
    uint64_t state = 0;
};
class ShadeInfo {
public:
    virtual ~ShadeInfo() = default;
    virtual float TraceShadowRay(Ray const& r, float t_max = BIGFLOAT) const = 0;
    virtual Color TraceSecondaryRay(Ray const& r, float& dist, bool recursive = true, bool isReflection = true) const = 0;
};

class ShadeInfoAdapter : public ShadeInfo {
public:
    // ShadeInfoAdapter.h:21 (the reference's signature)
    ShadeInfoAdapter(const std::vector<Light*>& lightList, const Scene* scenePtr, RNG& r) : lights(lightList), scene(scenePtr), rng(&r) {}
    // without a caller-owned RNG (the counter RNG needs none)
    ShadeInfoAdapter(const std::vector<Light*>& lightList, const Scene* scenePtr) : lights(lightList), scene(scenePtr) {}

    void SetHit(Ray const& r, HitInfo const& h) {
        hit = h;
        view = -r.dir.GetNormalized();
    }
    Vec3f P() const { return hit.p; }
    Vec3f N() const { return hit.N; }
    const Node* GetNode() const { return hit.node; }

    // ShadeInfoAdapter.h:49-88
    float TraceShadowRay(Ray const& r, float t_max = BIGFLOAT) const override {
        if (!scene) return 1.0f;
        const float Llen = r.dir.Length();
        if (Llen < 1e-12f) return 1.0f;
        const bool finiteSegment = (t_max < BIGFLOAT * 0.5f);
        const float segLenWorld = finiteSegment ? (Llen * t_max) : BIGFLOAT;
        const Vec3f Ldir = r.dir / Llen;
        const Vec3f Nn = this->N();
        const float nDotL = Nn % Ldir;
        const float baseBias = 1e-4f;
        const float nBias = baseBias * (1.0f + 2.0f * (1.0f - std::abs(nDotL)));
        const float dBias = baseBias * 0.5f;
        Ray sRay;
        sRay.p = P() + ((nDotL >= 0.0f) ? Nn : -Nn) * nBias + Ldir * dBias;
        sRay.dir = Ldir;
        HitInfo h;
        h.Init();
        h.z = BIGFLOAT;
        if (IntersectNodeRoot(sRay, h)) {
            const float dist = (h.p - sRay.p).Length();
            if (h.node == GetNode() && dist < (nBias + dBias) * 6.0f) return 1.0f;
            if (!finiteSegment) return dist > 0.0f ? 0.0f : 1.0f;
            const float allowance = nBias * 1.5f;
            return (dist < segLenWorld - allowance) ? 0.0f : 1.0f;
        }
        return 1.0f;
    }
    // ShadeInfoAdapter.h:90-92
    float TraceShadowRay(Vec3f const& dir, float t_max = BIGFLOAT) const { return TraceShadowRay(Ray(P(), dir), t_max); }

    // ShadeInfo state read by the secondary rays
    void SetPixel(int x, int y) { px = x; py = y; }
    void SetPixelSample(int s) { sampleIdx = s; }
    int X() const { return px; }
    int Y() const { return py; }
    int  GetDiffuseDepth() const { return diffuseDepth; }
    void SetDiffuseDepth(int d) { diffuseDepth = d; }
    bool CanBounce() const { return bounce < kMaxSpecularBounces; }   // :44
    void SetPhotonMapper(const PhotonMapBuilder* m) { mapper = m; }   // SetPhotonMap / SetCausticsMap + radii (:26-38)
    int bounce = 0;
    uint64_t seed = 0x9e3779b97f4a7c15ULL;   // key of the counter RNG (deviation 1)
    unsigned rngNode = 0;                    // this Shade() in the recursion tree (include/rtx.h, rtx_shade_point::node)
    int imageWidth = 0;                      // for the pixel index y*W+x of the RNG key

    // ShadeInfoAdapter.h:94-231. inRay.p is not read by the reference either (the ray starts at the biased P(), :111).
    // Both values of isReflection lead to the same result (:115-183); it selects the RNG node of the hit:
    // reflection = child 2, otherwise child 1.
    Color TraceSecondaryRay(Ray const& inRay, float& dist, bool recursive = true, bool isReflection = true) const override {
        dist = BIGFLOAT;
        if (!scene) return Color(0, 0, 0);
        if (!CanBounce()) return Color(0, 0, 0);
        GpuScene& gs = GetGpuScene(scene);
        rtx_shade_point at{};
        at.P[0] = hit.p.x; at.P[1] = hit.p.y; at.P[2] = hit.p.z;
        at.N[0] = hit.N.x; at.N[1] = hit.N.y; at.N[2] = hit.N.z;
        at.obj = gs.IndexOf(hit.node);
        at.bounce = bounce;
        at.diffuse_depth = diffuseDepth;
        at.pixel = py * imageWidth + px;
        at.sample = sampleIdx;
        at.node = rngNode;
        const float d[3] = {inRay.dir.x, inRay.dir.y, inRay.dir.z};
        float out[4] = {0, 0, 0, BIGFLOAT};
        if (rtx_trace_secondary(gs.Handle(), mapper ? mapper->Handle() : nullptr, &at, d, 1, recursive ? 1 : 0, isReflection ? 2 : 1,
                                seed, out) != RTX_OK) {
            std::cerr << "TraceSecondaryRay failed: " << rtx_last_error() << "\n";
            return Color(0, 0, 0);
        }
        dist = out[3];
        return Color(out[0], out[1], out[2]);
    }
    // ShadeInfoAdapter.h:233-235
    Color TraceSecondaryRay(Vec3f const& dir, float& dist, bool isReflection = true) const {
        return TraceSecondaryRay(Ray(P(), dir), dist, true, isReflection);
    }

private:
    bool IntersectNodeRoot(const Ray& ray, HitInfo& h) const {
        ExtendedHitInfo e;
        e.Init();
        e.z = h.z;
        GpuScene& gs = GetGpuScene(scene);
        rtx_ray r = detail::ToRay(ray, e.z);
        rtx_hit out;
        if (rtx_trace_closest(gs.Handle(), &r, 1, HIT_FRONT_AND_BACK, 0, &out) != RTX_OK || out.obj < 0) return false;
        detail::FillHit(gs, out, h);
        return true;
    }
    std::vector<Light*> lights;
    const Scene* scene;
    RNG* rng = nullptr;
    HitInfo hit;
    Vec3f view;
    const PhotonMapBuilder* mapper = nullptr;
    int px = 0, py = 0, sampleIdx = 0, diffuseDepth = 0;
    static constexpr int kMaxSpecularBounces = 5;   // :245
};

}  // namespace rtamd
