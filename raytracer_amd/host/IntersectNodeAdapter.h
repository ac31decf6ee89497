// Mirror of the reference's RayTracer/IntersectNodeAdapter.h: same names, arguments and bool results; the work is
// done by the extend kernel behind rtx_trace_closest (include/rtx.h). Batched variants are added because a GPU
// call per ray is only sensible for tests and tools; the renderer itself never calls these per ray (the whole path
// loop runs on the device).
#pragma once
#include "SceneAdapter.h"

namespace rtamd {

// IntersectNodeAdapter.h:13-20
struct ExtendedHitInfo : public HitInfo {
    const Light* hitLight = nullptr;
    void Init() {
        HitInfo::Init();
        hitLight = nullptr;
    }
};

namespace detail {
inline void FillHit(const GpuScene& gs, const rtx_hit& h, HitInfo& out) {
    out.z = h.t;
    out.node = gs.NodeOf(h.obj);
    out.p = Vec3f(h.p[0], h.p[1], h.p[2]);
    out.N = Vec3f(h.N[0], h.N[1], h.N[2]);
    out.GN = Vec3f(h.GN[0], h.GN[1], h.GN[2]);
    out.uvw = Vec3f(h.uvw[0], h.uvw[1], h.uvw[2]);
    out.dpdu = Vec3f(h.dpdu[0], h.dpdu[1], h.dpdu[2]);
    out.dpdv = Vec3f(h.dpdv[0], h.dpdv[1], h.dpdv[2]);
    out.duvw[0].Zero();
    out.duvw[1].Zero();
    out.mtlID = h.mtl_id;
    out.front = h.front != 0;
    out.light = h.light >= 0;
}
inline rtx_ray ToRay(const Ray& r, float tmax) {
    rtx_ray o{};
    o.o[0] = r.p.x; o.o[1] = r.p.y; o.o[2] = r.p.z;
    o.d[0] = r.dir.x; o.d[1] = r.dir.y; o.d[2] = r.dir.z;
    o.tmax = tmax;
    return o;
}
inline const Scene& EmptyScene() {
    static Scene s;
    return s;
}
}  // namespace detail

// IntersectNodeAdapter.h:100-108. outHit.z is in-out: the caller's upper bound on entry (:103).
inline bool IntersectNode(const Node* node, const Ray& inRay, HitInfo& outHit, int hitSide = HIT_FRONT) {
    GpuScene& gs = GetGpuScene(&detail::EmptyScene(), node);
    rtx_ray r = detail::ToRay(inRay, outHit.z);
    rtx_hit h;
    if (rtx_trace_closest(gs.Handle(), &r, 1, hitSide, 0, &h) != RTX_OK) return false;   // errors read as "no hit" (§8b)
    if (h.obj < 0) return false;
    detail::FillHit(gs, h, outHit);
    return true;
}

// IntersectNodeAdapter.h:111-151
inline bool IntersectSceneWithLights(const Scene* scene, const Ray& ray, ExtendedHitInfo& outHit, int hitSide = HIT_FRONT) {
    GpuScene& gs = GetGpuScene(scene);
    rtx_ray r = detail::ToRay(ray, outHit.z);
    rtx_hit h;
    if (rtx_trace_closest(gs.Handle(), &r, 1, hitSide, 1, &h) != RTX_OK) return false;
    if (h.obj < 0 && h.light < 0) return false;
    detail::FillHit(gs, h, outHit);
    outHit.hitLight = h.light >= 0 ? gs.Lights()[h.light] : nullptr;
    return true;
}

// Batched form of IntersectSceneWithLights / IntersectNode(&scene->rootNode, ...): hits[i].z carries the bound in.
inline void IntersectSceneBatch(const Scene* scene, const std::vector<Ray>& rays, std::vector<ExtendedHitInfo>& hits, int hitSide,
                                bool withLights, std::vector<bool>* hitFlags = nullptr) {
    GpuScene& gs = GetGpuScene(scene);
    std::vector<rtx_ray> rr(rays.size());
    std::vector<rtx_hit> hh(rays.size());
    hits.resize(rays.size());
    for (size_t i = 0; i < rays.size(); ++i) rr[i] = detail::ToRay(rays[i], hits[i].z);
    if (hitFlags) hitFlags->assign(rays.size(), false);
    if (rtx_trace_closest(gs.Handle(), rr.data(), rr.size(), hitSide, withLights ? 1 : 0, hh.data()) != RTX_OK) return;
    for (size_t i = 0; i < rays.size(); ++i) {
        const bool hit = hh[i].obj >= 0 || hh[i].light >= 0;
        if (hit) {
            detail::FillHit(gs, hh[i], hits[i]);
            hits[i].hitLight = hh[i].light >= 0 ? gs.Lights()[hh[i].light] : nullptr;
        }
        if (hitFlags) (*hitFlags)[i] = hit;
    }
}

}  // namespace rtamd
