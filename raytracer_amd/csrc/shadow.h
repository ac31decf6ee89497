// Device-side shadow query: Visible() / VisibleSimple() of the reference (RayTracer/RayCaster.cpp:178-278).
// One lane runs the whole chain: up to 8 closest-hit hops through transmissive surfaces, each attenuated by
// (1 - Schlick Fresnel) * max(kt); an opaque surface ends the chain with 0.
#pragma once
#include "surface.h"

// RayCaster.cpp:108-112
__device__ __forceinline__ f3 offset_point(f3 p, f3 n, f3 dir) {
    const f3 nn = safe_normalize3(n);
    const float sign = dot3(dir, nn) >= 0.0f ? 1.0f : -1.0f;
    return p + nn * (1e-4f * sign);
}

// Geometric normal + front flag of a triangle hit (objects.cpp:175-177,190) without the rest of HitInfo.
template <class SC>
__device__ __forceinline__ void triangle_gn(const SC& sc, const RayT& r, int slot, float t, f3& GN, bool& front, int& obj) {
    const float4* tris4 = reinterpret_cast<const float4*>(sc.tris);
    const float4 b = tris4[(size_t)slot * 3 + 1];
    const float4 c = tris4[(size_t)slot * 3 + 2];
    if (SC::kXF && __float_as_uint(b.w) != 0u) {   // Sphere / Plane / mesh under a node transform: through the object's space
        SurfHit sh;
        resolve_triangle(sc, r, HitT{t, 0.0f, 0.0f, slot}, sh);
        GN = sh.GN;
        front = sh.front;
        obj = sh.obj;
        return;
    }
    GN = cross3(mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z));
    const float l2 = len2_3(GN);
    if (l2 > 0.0f) GN = GN / sqrtf(l2);
    else GN = mk3(0.0f, 0.0f, 1.0f);
    front = dot3(mk3(r.dx, r.dy, r.dz), GN) < 0.0f;
    obj = (int)(sc.attrs[slot].obj_mtl & 0xFFFFu);
}

// End of one hop of the chain (RayCaster.cpp:205-276): h is the closest hit of the segment (o, d, maxDist), lights
// included. Returns true when the chain ends (value in `result`), false when the next hop starts from the updated
// (o, maxDist) with the updated running visibility.
// word: the hit record's `obj` word when the caller has kept it from the triangle test (TravState::bestWord), else 0. TRI_OPAQUE
// in it is the builder's statement of the two conditions below that end the chain with 0 - no material, or max(kt) <= 0
// (bvh_build.cpp emit_tri) - and saves the three dependent fetches (triangle, object, material) that would find them out again.
template <class SC>
__device__ __forceinline__ bool shadow_hop_end(const SC& sc, f3& o, f3 d, float& maxDist, const HitT& h, bool useTransmittance,
                                               int& hop, float& visibility, float& result, uint32_t word = 0u) {
    result = visibility;
    if (h.slot == -1) return true;   // nothing blocking
    if (h.slot <= -2) return true;   // hit the light itself
    f3 hp = o + d * h.t;
    if (SC::kXF && sc.tris[h.slot].kind != 0u) {   // HitInfo::p comes back through the node transform (IntersectNodeAdapter.h:93); lean scenes hold no such record
        SurfHit sh;
        resolve_triangle(sc, RayT{o.x, o.y, o.z, d.x, d.y, d.z, maxDist}, HitT{h.t, 0.0f, 0.0f, h.slot}, sh);
        hp = sh.p;
    }
    const float dist = len3(hp - o);
    if (dist >= maxDist - 1e-4f) return true;
    result = 0.0f;
    if (!useTransmittance) return true;    // VisibleSimple :189
    if ((word & TRI_OPAQUE) != 0u) return true;
    f3 GN;
    bool front;
    int obj;
    const RayT r{o.x, o.y, o.z, d.x, d.y, d.z, maxDist};
    triangle_gn(sc, r, h.slot, h.t, GN, front, obj);
    const int mi = sc.objs[obj].material;
    if (mi < 0) return true;
    const DevMaterial& m = sc.mtls[mi];
    const float transFactor = fmaxf(m.kt[0], fmaxf(m.kt[1], m.kt[2]));
    if (transFactor <= 0.0f) return true;
    const float ior = m.ior;
    float cosTheta = fabsf(dot3(GN, -d));
    const float etaI = front ? 1.0f : ior;
    const float etaT = front ? ior : 1.0f;
    cosTheta = fminf(fmaxf(cosTheta, 0.0f), 1.0f);
    float r0 = (etaI - etaT) / (etaI + etaT);
    r0 = r0 * r0;
    const float F = r0 + (1.0f - r0) * rtx_powf(1.0f - cosTheta, 5.0f);
    const float transmit = fmaxf(0.0f, fminf(1.0f, (1.0f - F) * transFactor));
    visibility *= fmaxf(0.0f, fminf(1.0f, transmit));
    if (visibility <= 1e-4f) return true;
    o = offset_point(hp, GN, d);
    maxDist -= dist;
    if (maxDist <= 1e-4f) {
        result = visibility;
        return true;
    }
    ++hop;
    return hop >= 8;   // kMaxHops with transmittance: the chain gives up with 0
}

// Early end of a shadow hop: the hit just accepted (s.best) is a blocker that ends the chain with 0 if it turns out to be
// the hop's closest hit - an opaque surface (any surface for VisibleSimple) in front of the segment's end and of every
// renderable light. A closer OPAQUE hit gives 0 as well; a closer hit that lets light through does not (its next hop
// starts from an offset origin and need not meet this blocker again), so the caller goes on looking for those alone
// (shadow_kernel). The margins keep hits within rounding of the segment end or of a light surface on the full path.
__device__ __forceinline__ bool shadow_hit_decides(const DevScene& sc, const TravState& s, float maxDist, bool useTransmittance) {
    if (s.best.slot < 0) return false;
    if (!(s.best.t < (maxDist - 1e-4f) * 0.99999f)) return false;
    if (useTransmittance && (s.bestWord & TRI_OPAQUE) == 0u) return false;   // (the record's word, kept from the triangle test)
    for (int i = 0; i < sc.n_lights; ++i) {
        const DevLight& L = sc.lights[i];
        if (L.size <= 0.0f) continue;
        // conservative: a light whose sphere the ray enters before (or near) the hit keeps the full path
        const float ocx = s.r.ox - L.pos[0], ocy = s.r.oy - L.pos[1], ocz = s.r.oz - L.pos[2];
        const float b = ocx * s.r.dx + ocy * s.r.dy + ocz * s.r.dz;
        const float c = (ocx * ocx + ocy * ocy + ocz * ocz) - L.size * L.size;
        const float disc = b * b - c;
        if (disc < 0.0f) continue;
        const float root = sqrtf(disc);
        if (-b + root < 0.0f) continue;                       // sphere wholly behind the origin
        if (-b - root <= s.best.t * 1.00001f) return false;   // entered (or origin inside) before the hit
    }
    return true;
}

// o = OffsetPoint(origin, normal, dir), d = SafeNormalize(dir) already applied (RayCaster.cpp:199)
template <bool COUNT, class SC>
__device__ __forceinline__ float visible_chain_from(const SC& sc, f3 o, f3 d, float maxDist, bool useTransmittance,
                                                    TravStack st, uint32_t* hops, uint32_t* nodeCount, uint32_t* triCount) {
    float visibility = 1.0f, result = 1.0f;
    int hop = 0;
    for (;;) {
        const RayT r{o.x, o.y, o.z, d.x, d.y, d.z, maxDist};
        (*hops)++;
        HitT h = traverse_closest<COUNT>(sc, r, RTX_HIT_FRONT_AND_BACK, st, nodeCount, triCount);
        intersect_lights(sc, r, h);
        if (shadow_hop_end(sc, o, d, maxDist, h, useTransmittance, hop, visibility, result)) return result;
    }
}

template <bool COUNT, class SC>
__device__ __forceinline__ float visible_chain(const SC& sc, f3 origin, f3 normal, f3 dirIn, float maxDist,
                                               bool useTransmittance, TravStack st, uint32_t* hops, uint32_t* nodeCount,
                                               uint32_t* triCount) {
    return visible_chain_from<COUNT>(sc, offset_point(origin, normal, dirIn), safe_normalize3(dirIn), maxDist, useTransmittance,
                                     st, hops, nodeCount, triCount);
}
