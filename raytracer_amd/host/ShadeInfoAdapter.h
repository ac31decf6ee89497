// Mirror of the shadow query of the reference's RayTracer/ShadeInfoAdapter.h (legacy shading mode,
// SHADING_USE_LEGACY=1): TraceShadowRay with its bias rules (:49-92), evaluated with one closest-hit on the GPU.
// TraceSecondaryRay (:94-231) recurses into the legacy Shade() functions of Shading_Legacy.cpp, which are outside
// this round's scope; it is declared and reports "not available".
#pragma once
#include <cmath>
#include "IntersectNodeAdapter.h"

namespace rtamd {

class ShadeInfoAdapter {
public:
    ShadeInfoAdapter(const std::vector<Light*>& lightList, const Scene* scenePtr) : lights(lightList), scene(scenePtr) {}

    void SetHit(Ray const& r, HitInfo const& h) {
        hit = h;
        view = -r.dir.GetNormalized();
    }
    Vec3f P() const { return hit.p; }
    Vec3f N() const { return hit.N; }
    const Node* GetNode() const { return hit.node; }

    // ShadeInfoAdapter.h:49-88
    float TraceShadowRay(Ray const& r, float t_max = BIGFLOAT) const {
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

    // ShadeInfoAdapter.h:94-231 — legacy recursion, not part of this round (see header comment)
    Color TraceSecondaryRay(Ray const&, float& dist, bool = true, bool = true) const {
        dist = BIGFLOAT;
        return Color(0, 0, 0);
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
    HitInfo hit;
    Vec3f view;
};

}  // namespace rtamd
