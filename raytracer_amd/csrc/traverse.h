// Device-side closest-hit traversal of the flattened scene (gfx950, wave64).
//
// Semantics follow the reference's closest-hit: Box::IntersectRay slab culls (RayTracer/scene_box.cpp:8-46)
// around TriObj::IntersectTriangle (RayTracer/objects.cpp:140-171, Moller-Trumbore with EPS = 1e-7, inclusive
// barycentric edges, accept 0 < EPS < t < current z). The reference recurses over a pointer BVH2 and always
// visits both children (RayTracer/triobj_bvh.cpp:27-31); here each lane walks the quantised 4-wide BVH of
// layout.h iteratively, nearest child first, with a short per-lane stack in LDS (8-byte entries
// {child, entry distance}) that spills to a global slab when deeper than STACK_LDS.
//
// Exact-t ties: the reference keeps whichever triangle its scene/BVH order visits first (strict '<'); here the
// smaller global triangle id wins, which is the order of the reference's brute-force path (objects.cpp:257-264).
#pragma once
#include <hip/hip_runtime.h>
#include "layout.h"

#define RTX_EPS 1e-7f
#define RTX_BIG 1.0e30f

struct RayT {
    float ox, oy, oz, dx, dy, dz, tmax;
};
struct HitT {
    float t, u, v;
    int slot;   // index into tris/attrs (BVH order); -1 = miss; -2-k = light k
};

// Per-lane traversal stack. LDS part is laid out [entry][thread] so that a wave's ds_read_b64/ds_write_b64 are
// conflict-free; the spill part is [entry][global thread] or [global thread][entry] (make_trav_stack).
struct TravStack {
    uint2* lds;          // &stack[0][threadIdx.x]
    uint2* spill;        // &slab[0 * stride + gtid]
    uint32_t stride;     // total threads of the grid
    __device__ __forceinline__ void push(int& sp, uint32_t ref, float t) {
        uint2 e = make_uint2(ref, __float_as_uint(t));
        typedef __attribute__((address_space(3))) unsigned long long* LdsPtrW;   // (LDS by type, as in pop: no flat store through a selected pointer)
        if (sp < STACK_LDS) ((LdsPtrW)lds)[sp * TRAV_BLOCK] = (unsigned long long)e.x | ((unsigned long long)e.y << 32);
        else if (sp < STACK_LDS + STACK_GLOBAL) spill[(size_t)(sp - STACK_LDS) * stride] = e;
        // deeper than STACK_LDS+STACK_GLOBAL cannot happen: host rejects BVHs with 3*depth+1 above the budget
        ++sp;
    }
    // (the LDS entry is read whatever the depth, at a clamped index: written as "LDS or slab" the two loads become ONE flat load
    // through a selected generic pointer - every pop then goes the flat path and waits for the vector-memory and the LDS counters)
    __device__ __forceinline__ uint2 pop(int& sp) {
        --sp;
        typedef __attribute__((address_space(3))) const unsigned long long* LdsPtr;   // (an LDS pointer by type: it cannot be merged with the slab's)
        const unsigned long long w = ((LdsPtr)lds)[(sp < STACK_LDS ? sp : STACK_LDS - 1) * TRAV_BLOCK];
        uint2 e = make_uint2((uint32_t)w, (uint32_t)(w >> 32));
        if (sp >= STACK_LDS) e = spill[(size_t)(sp - STACK_LDS) * stride];
        return e;
    }
};

// the stack of lane gtid of a grid of nThreads lanes; ldsLane = &stack[0][threadIdx.x]. LANE_MAJOR: the spilled part is
// [global thread][entry] - a lane's entries share 64-byte lines - instead of [entry][global thread]. Measured per kernel
// (C3): extend 331.4 -> 327.6 ms and the legacy kernel 481 -> 467 ms lane-major, shadow 161.7 -> 170.9 ms (its short
// traversals seldom go deeper than one spilled entry, which the interleaved rows serve to a whole wave at once).
template <bool LANE_MAJOR = false>
__device__ __forceinline__ TravStack make_trav_stack(uint2* ldsLane, uint2* slab, uint32_t gtid, uint32_t nThreads) {
    if (LANE_MAJOR) return TravStack{ldsLane, slab + (size_t)gtid * STACK_GLOBAL, 1u};
    return TravStack{ldsLane, slab + gtid, nThreads};
}

__device__ __forceinline__ float scale_from_exp(uint32_t e) { return __uint_as_float(e << 23); }

// Box::IntersectRay (scene_box.cpp:8-46) as written. The reference reaches an object's IntersectRay only through this test
// on the object's own box, with the ray in node coordinates (IntersectNodeAdapter.h:63-75), and the whole scene through
// the same test on the root's child box (:30-32). The test is exact arithmetic on an exact box, so a hit within rounding
// of the box's silhouette - an edge of a Plane or of a box-shaped mesh, the rim of a floor quad - is culled or not by the
// rounding of THESE operations. They are repeated by the general kernels (SC::kXF: scenes that hold a mesh under a node
// transform, a Sphere or a Plane): in test_analytic on the way in, and in trav_verify for the mesh hit a traversal ends
// with. The lean kernels - scenes of world-space triangles only, every object of cerealBar.xml - do NOT repeat them
// (DESIGN.md 4 "box culls"): wherever the test sits, it costs extend_kernel its registers (+78 % where a hit is
// accepted, +34 % where a traversal ends; a one-register "skip this object" compare alone +18 %). box = {min xyz, max xyz}.
__device__ __forceinline__ bool ref_box_hit(const float* box, float px, float py, float pz, float dx, float dy, float dz, float t_max) {
#pragma clang fp contract(off)
    const float p[3] = {px, py, pz}, d[3] = {dx, dy, dz};
    float tmin = 0.0f, tmax = t_max;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float minv = box[i], maxv = box[3 + i];
        if (fabsf(d[i]) < 1e-20f) {
            if (p[i] < minv || p[i] > maxv) return false;
            continue;
        }
        const float invD = 1.0f / d[i];
        float t0 = (minv - p[i]) * invD, t1 = (maxv - p[i]) * invD;
        if (t0 > t1) {
            const float tt = t0;
            t0 = t1;
            t1 = tt;
        }
        if (t0 > tmin) tmin = t0;
        if (t1 < tmax) tmax = t1;
        if (tmax < tmin) return false;
    }
    return tmax >= 0.0f && tmin <= t_max;
}

// Is point p (the hit, a distance t along a ray) inside `box` by a margin that is large against the rounding of
// ref_box_hit - 0.1 % of the box's extent plus 1e-5 of the magnitudes involved, per axis? Then every slab is entered
// before and left after the hit by far more than an ulp of the slab distances, and ref_box_hit accepts; no division.
__device__ __forceinline__ bool well_inside_box(const float* box, float px, float py, float pz, float t) {
    const float q[3] = {px, py, pz};
    bool in = true;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float eps = 1e-3f * (box[3 + i] - box[i]) + 1e-5f * (fabsf(q[i]) + fabsf(t));
        in = in && q[i] > box[i] + eps && q[i] < box[3 + i] - eps;
    }
    return in;
}
// (out of line for the shadow kernel, whose loop then keeps its registers: 221 -> 191 ms per C3 frame; a call inside the
// shade kernel costs it 100 ms instead; reached for hits near a box face only)
// (the boxes by value: a pointer into the kernel's scene argument would make the compiler keep a copy of the whole descriptor in
// scratch memory for the sake of this one call)
struct Box6 {
    float v[6];
};
__device__ __noinline__ bool ref_reaches_exact(Box6 rootBox, Box6 objBox, float ox, float oy, float oz, float dx, float dy, float dz, float tmax) {
    if (!ref_box_hit(rootBox.v, ox, oy, oz, dx, dy, dz, tmax)) return false;
    return ref_box_hit(objBox.v, ox, oy, oz, dx, dy, dz, tmax);
}
// Does the reference reach triangle `slot` of a scene of world-space triangles with ray r, which hits it at t? (the
// root's box, then the object's own: IntersectNodeAdapter.h:30-32, :63-75). The exact tests run only for hits near a
// face of one of the two boxes (flat objects, rims, the walls of the scene).
template <bool OUTLINE>
__device__ __forceinline__ bool ref_reaches_box(const DevScene& sc, const RayT& r, const float* obox, float t) {
    const float px = r.ox + r.dx * t, py = r.oy + r.dy * t, pz = r.oz + r.dz * t;
    const bool sure = t < 1e6f && r.tmax > t * 1.001f && well_inside_box(sc.bounds, px, py, pz, t) && well_inside_box(obox, px, py, pz, t);
    if (sure) return true;
    if (OUTLINE) {
        Box6 rb, ob;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            rb.v[i] = sc.bounds[i];
            ob.v[i] = obox[i];
        }
        return ref_reaches_exact(rb, ob, r.ox, r.oy, r.oz, r.dx, r.dy, r.dz, r.tmax);
    }
    if (!ref_box_hit(sc.bounds, r.ox, r.oy, r.oz, r.dx, r.dy, r.dz, r.tmax)) return false;
    return ref_box_hit(obox, r.ox, r.oy, r.oz, r.dx, r.dy, r.dz, r.tmax);
}
// (word: the triangle record's `obj` word)
__device__ __forceinline__ bool ref_reaches_interior(uint32_t word, const RayT& r, float t) {
    return (word & TRI_INTERIOR) != 0u && t < TRI_INTERIOR_TMAX && r.tmax > t * 1.001f;   // implies `sure` of ref_reaches_box
}
template <bool OUTLINE>
__device__ __forceinline__ bool ref_reaches_triangle(const DevScene& sc, const RayT& r, int slot, float t) {
    const uint32_t word = sc.tris[slot].obj;
    if (ref_reaches_interior(word, r, t)) return true;   // no box to fetch: most triangles of most scenes
    const uint32_t obj = word & TRI_OBJ_MASK;
    float ob[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) ob[i] = sc.objXf[obj].box[i];
    return ref_reaches_box<OUTLINE>(sc, r, ob, t);
}

// One triangle, objects.cpp:140-171 and :190-191. The arithmetic order is the reference's; contraction is off so
// that u, v and t are bit-identical to an fp32 CPU evaluation. r is the ray in the triangle's own space.
// bestWord: the accepted record's `obj` word (TRI_OPAQUE, TRI_INTERIOR), already in a register here - the shadow kernel decides
// with it what would otherwise be a dependent fetch per accepted hit and per hop end; kernels that never read it drop it.
__device__ __forceinline__ void test_triangle(const float4 a, const float4 b, const float4 c, const RayT& r, int hitSide,
                                              uint32_t slot, const TriRec* tris, HitT& best, uint32_t& bestPrim, uint32_t& bestWord,
                                              uint32_t excl = 0x7FFFFFFFu) {
#pragma clang fp contract(off)
    const float e1x = b.x, e1y = b.y, e1z = b.z;
    const float e2x = c.x, e2y = c.y, e2z = c.z;
    // pvec = dir ^ e2
    const float px = r.dy * e2z - r.dz * e2y;
    const float py = r.dz * e2x - r.dx * e2z;
    const float pz = r.dx * e2y - r.dy * e2x;
    const float det = e1x * px + e1y * py + e1z * pz;
    if (hitSide == RTX_HIT_FRONT && det <= RTX_EPS) return;
    if (hitSide == RTX_HIT_BACK && det >= -RTX_EPS) return;
    if (fabsf(det) < RTX_EPS) return;
    const float invDet = 1.0f / det;
    const float tx = r.ox - a.x, ty = r.oy - a.y, tz = r.oz - a.z;
    const float u = (tx * px + ty * py + tz * pz) * invDet;
    if (u < 0.0f || u > 1.0f) return;
    // qvec = tvec ^ e1
    const float qx = ty * e1z - tz * e1y;
    const float qy = tz * e1x - tx * e1z;
    const float qz = tx * e1y - ty * e1x;
    const float v = (r.dx * qx + r.dy * qy + r.dz * qz) * invDet;
    if (v < 0.0f || u + v > 1.0f) return;
    const float t = (e2x * qx + e2y * qy + e2z * qz) * invDet;
    if (t <= RTX_EPS) return;
    const uint32_t prim = __float_as_uint(a.w);
    if (t > best.t || (t == best.t && (best.slot < 0 || prim >= bestPrim))) return;  // objects.cpp:170 (+ tie rule)
    if (hitSide != RTX_HIT_FRONT_AND_BACK) {
        // objects.cpp:175-177,190-191: side test on the normalised geometric normal
        float gx = e1y * e2z - e1z * e2y;
        float gy = e1z * e2x - e1x * e2z;
        float gz = e1x * e2y - e1y * e2x;
        const float l2 = gx * gx + gy * gy + gz * gz;
        if (l2 > 0.0f) {
            const float l = sqrtf(l2);
            gx = gx / l; gy = gy / l; gz = gz / l;
        } else {
            gx = 0.0f; gy = 0.0f; gz = 1.0f;
        }
        const bool front = (r.dx * gx + r.dy * gy + r.dz * gz) < 0.0f;
        if ((hitSide == RTX_HIT_FRONT && !front) || (hitSide == RTX_HIT_BACK && front)) return;
    }
    if ((__float_as_uint(c.w) & TRI_OBJ_MASK) == excl) return;   // general kernels only (trav_verify); folds away when excl is the default
    best.t = t;
    best.u = u;
    best.v = v;
    best.slot = (int)slot;
    bestPrim = prim;
    bestWord = __float_as_uint(c.w);
}

// Analytic objects (one record per object, TriRec::kind != 0): the ray goes to object space with the node's inverse
// transform, direction not renormalised, so t is the same parameter in both spaces (Node::ToNodeCoords,
// IntersectNodeAdapter.h:62). Sphere::IntersectRay objects.cpp:21-51, Plane::IntersectRay objects.cpp:89-104.
__device__ __forceinline__ void ray_to_object(const DevScene& sc, uint32_t obj, const RayT& r, float p[3], float d[3]) {
    const float* M = sc.objXf[obj].inv;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        p[k] = M[4 * k] * r.ox + M[4 * k + 1] * r.oy + M[4 * k + 2] * r.oz + M[4 * k + 3];
        d[k] = M[4 * k] * r.dx + M[4 * k + 1] * r.dy + M[4 * k + 2] * r.dz;
    }
}
__device__ __forceinline__ void test_analytic(const DevScene& sc, uint32_t kind, uint32_t word, uint32_t prim, const RayT& r,
                                              int hitSide, uint32_t slot, HitT& best, uint32_t& bestPrim, uint32_t& bestWord) {
#pragma clang fp contract(off)
    const uint32_t obj = word & TRI_OBJ_MASK;
    float p[3], d[3];
    ray_to_object(sc, obj, r, p, d);
    if (!ref_box_hit(sc.objXf[obj].box, p[0], p[1], p[2], d[0], d[1], d[2], best.t)) return;   // [-1,1]^3, or the flat [-1,1]^2 x {0}
    float t;
    bool front;
    if (kind == RTX_OBJ_SPHERE) {
        const float A = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
        const float B = 2.0f * (p[0] * d[0] + p[1] * d[1] + p[2] * d[2]);
        const float C = (p[0] * p[0] + p[1] * p[1] + p[2] * p[2]) - 1.0f;
        const float disc = B * B - 4 * A * C;
        if (disc < 0.0f) return;
        const float s = sqrtf(disc);
        const float inv2A = 0.5f / A;
        const float t0 = (-B - s) * inv2A;
        const float t1 = (-B + s) * inv2A;
        t = -1.0f;
        if (t0 > RTX_EPS) t = t0;
        else if (t1 > RTX_EPS) t = t1;
        if (t < 0.0f || t > best.t) return;
        const float hx = p[0] + d[0] * t, hy = p[1] + d[1] * t, hz = p[2] + d[2] * t;
        const float len = sqrtf(hx * hx + hy * hy + hz * hz);
        front = (d[0] * (hx / len) + d[1] * (hy / len) + d[2] * (hz / len)) < 0.0f;
    } else {
        const float denom = d[2];
        if (fabsf(denom) < RTX_EPS) return;
        t = -p[2] / denom;
        if (t <= RTX_EPS || t > best.t) return;
        front = denom < 0.0f;
    }
    if ((hitSide == RTX_HIT_FRONT && !front) || (hitSide == RTX_HIT_BACK && front)) return;
    if (kind == RTX_OBJ_PLANE) {
        const float hx = p[0] + d[0] * t, hy = p[1] + d[1] * t;
        if (hx < -1.0f || hx > 1.0f) return;
        if (hy < -1.0f || hy > 1.0f) return;
    }
    if (t == best.t && (best.slot < 0 || prim >= bestPrim)) return;   // t >= hInfo.z rejects; ties: smaller id wins
    best.t = t;
    best.u = 0.0f;
    best.v = 0.0f;
    best.slot = (int)slot;
    bestPrim = prim;
    bestWord = word;
}
// Scene argument of the traversal functions with one compile-time fact attached: whether the scene holds meshes under a
// non-identity node transform (RTX_KIND_XFTRI records). Their test is a second inlined copy of the triangle test behind
// a ray transform; merely compiling it into the traversal loop raises its register pressure (28 spilled VGPRs, extend
// +68 % on scenes that hold no such mesh - every object of cerealBar.xml is an identity node), and as a non-inlined
// call it costs more still. So the kernels exist in two instantiations and the launch wrappers pick by DevScene::has_xf.
// The Sphere / Plane test rides in the same instantiation (with its reference box test it costs the lean kernel 18 %):
// kXF = false is a scene of world-space triangles and nothing else - every object of cerealBar.xml.
template <bool XF>
struct DevSceneT : DevScene {
    static constexpr bool kXF = XF;
    __device__ __forceinline__ explicit DevSceneT(const DevScene& s) : DevScene(s) {}
};
template <class SC>
__device__ __forceinline__ void test_primitive(const SC& sc, const float4 a, const float4 b, const float4 c, const RayT& r,
                                               int hitSide, uint32_t slot, HitT& best, uint32_t& bestPrim, uint32_t& bestWord, uint32_t excl) {
    const uint32_t kind = __float_as_uint(b.w);
    if (!SC::kXF) excl = 0x7FFFFFFFu;   // (a constant there: the compare in test_triangle can never hold and goes away)
    if (kind == 0u) test_triangle(a, b, c, r, hitSide, slot, sc.tris, best, bestPrim, bestWord, excl);
    else if (!SC::kXF) return;   // identity meshes only (DevScene::has_xf == 0): no other kind of record exists
    else if (kind == RTX_KIND_XFTRI) {
        // a mesh under a non-identity node transform: the ray goes to node coordinates (direction not renormalised, so t
        // is the same parameter) and meets the object's own fp32 vertices (IntersectNodeAdapter.h:62 -> objects.cpp:140)
        float p[3], d[3];
        ray_to_object(sc, __float_as_uint(c.w) & TRI_OBJ_MASK, r, p, d);
        test_triangle(a, b, c, RayT{p[0], p[1], p[2], d[0], d[1], d[2], r.tmax}, hitSide, slot, sc.tris, best, bestPrim, bestWord, excl);
    } else
        test_analytic(sc, kind, __float_as_uint(c.w), __float_as_uint(a.w), r, hitSide, slot, best, bestPrim, bestWord);
}

// ---- traversal as a small state machine -------------------------------------------------------------------
// cur: RTX_TRAV_DONE, an inner-node index, or a leaf reference. The kernels run it "while-while": all lanes of a
// wave first descend inner nodes until each holds a leaf (or is done), then all test their leaves together, so
// the node code and the triangle code each run with as many active lanes as possible.
#define RTX_TRAV_DONE 0xFFFFFFFFu   // == RTX_EMPTY_CHILD: never a valid child to visit
#define RTX_LANE_IDLE 0xFFFFFFFEu   // (kernels' lane state, never set by the traversal: "this lane holds no ray"; no leaf reference ends
                                    //  in 0b110 - a leaf holds at most RTX_MAX_LEAF <= 4 records)

struct TravState {
    RayT r;
    float idx, idy, idz;     // 1 / dir (IEEE: +-inf for axis-parallel rays)
    HitT best;
    uint32_t bestPrim;
    uint32_t bestWord;       // TriRec::obj of the best hit (0 while there is none)
    uint32_t cur;
    int sp;
    uint32_t excl;           // general kernels: object left out of this traversal (its box culled it for this ray: trav_verify)
};
#define RTX_NO_OBJECT 0x7FFFFFFFu

__device__ __forceinline__ void trav_init(TravState& s, const DevScene& sc, const RayT& r) {
    s.r = r;
    // 1 / dir serves the slab tests alone - a box is entered with a margin of 1e-5 (RTX_CULL_SLACK), hits are decided by the
    // triangle test, which does not read it - so the hardware reciprocal (1 ulp; +-0 -> +-inf) can stand in for three IEEE
    // divisions. kernels_trace.hip asks for it (extend -1.5 ms per C3 frame); the shadow kernel sits on a register cliff where
    // the other instruction sequence costs it 16 more spilled registers (171 -> 200 ms), so it is not the default.
#ifdef RTX_TRAV_FAST_RCP
    s.idx = __builtin_amdgcn_rcpf(r.dx);
    s.idy = __builtin_amdgcn_rcpf(r.dy);
    s.idz = __builtin_amdgcn_rcpf(r.dz);
#else
    s.idx = 1.0f / r.dx;
    s.idy = 1.0f / r.dy;
    s.idz = 1.0f / r.dz;
#endif
    // ... and kept finite: an axis-parallel ray (1/0 = +-inf) is as parallel at +-2^60, and the slab test can then form a plane's
    // distance with ONE fused multiply-add, q * (scale/dir) + (origin - ro)/dir, without meeting inf - inf (trav_inner_step)
#ifndef RTX_TRAV_SLAB_TWO_OPS
    s.idx = __builtin_amdgcn_fmed3f(s.idx, -0x1p60f, 0x1p60f);
    s.idy = __builtin_amdgcn_fmed3f(s.idy, -0x1p60f, 0x1p60f);
    s.idz = __builtin_amdgcn_fmed3f(s.idz, -0x1p60f, 0x1p60f);
#endif
    s.best.t = r.tmax;      // HitInfo::z in-out upper bound (IntersectNodeAdapter.h:28)
    s.best.u = 0.0f;
    s.best.v = 0.0f;
    s.best.slot = -1;
    s.bestPrim = 0xFFFFFFFFu;
    s.bestWord = 0u;
    s.cur = sc.root;        // RTX_EMPTY_CHILD for an empty scene == DONE
    s.sp = 0;
    s.excl = RTX_NO_OBJECT;
    // NaN or infinite origin / direction, NaN limit: no hit, as the reference's box test at the root decides for such a
    // ray (every comparison of scene_box.cpp:8-46 fails). The slab test here drops NaNs instead, so it has to be said.
    const float probe = (r.ox + r.oy + r.oz) + (r.dx + r.dy + r.dz);
    if (!(fabsf(probe) < __builtin_inff()) || r.tmax != r.tmax) s.cur = RTX_TRAV_DONE;
}
__device__ __forceinline__ bool trav_is_inner(uint32_t cur) { return (cur & RTX_LEAF_FLAG) == 0u; }

// A box is entered while tnear <= tfar * (1 + 1e-5). tnear of a flat box around a triangle at distance t is t itself
// up to the rounding of (q - ro) * (1/dir), a few ulp either way: with a narrower margin the second of two faces at
// exactly the same t (duplicates, coplanar overlaps) is culled half of the time once the first has set best.t = t, and
// the tie rule (smaller face id) would depend on the visiting order. Ten-odd ulp of t in extra visits cost nothing.
#ifndef RTX_CULL_SLACK
#define RTX_CULL_SLACK 1.00001f
#endif
// next subtree that can still hold a closer (or tying) hit
__device__ __forceinline__ void trav_pop(TravState& s, TravStack& st) {
    uint32_t next = RTX_TRAV_DONE;
    while (s.sp > 0) {
        const uint2 e = st.pop(s.sp);
        if (__uint_as_float(e.y) <= s.best.t * RTX_CULL_SLACK) {
            next = e.x;
            break;
        }
    }
    s.cur = next;
}
// Child references carry RTX_REF_TRANS when the subtree holds a surface that lets light through (set by the builder), and so
// do s.cur and the stack entries; the bit is masked where a node is addressed and where a leaf reference is decoded.
// trav_skip_opaque drops everything that does not carry it, without a fetch: what is left of a shadow hop once its
// blocker is known (shadow_kernel).
#define RTX_REF_TRANS 0x40000000u   // free in every reference: fewer than 2^27 triangles and nodes (bvh_build.cpp)
__device__ __forceinline__ void trav_skip_opaque(TravState& s, TravStack& st) {
    if (s.cur == RTX_TRAV_DONE || (s.cur & RTX_REF_TRANS) != 0u) return;
    uint32_t next = RTX_TRAV_DONE;
    while (s.sp > 0) {
        const uint2 e = st.pop(s.sp);
        if ((e.x & RTX_REF_TRANS) != 0u && __uint_as_float(e.y) <= s.best.t * RTX_CULL_SLACK) {
            next = e.x;
            break;
        }
    }
    s.cur = next;
}

// One inner node (64 B = 4 x dwordx4). The slab test works on the quantised planes directly: with scale = 2^e exact, the
// distance to plane q of an axis is q * (scale/dir) + (o - ro)/dir - one fused multiply-add per plane, two planes per
// v_pk_fma_f32 (round 3; before: a packed subtract and a packed multiply in the node's quantised frame). Which plane is the
// near one follows from the sign of the direction, so the near/far bytes of all four children are picked with one select per
// axis before they are converted (byte -> float is a single v_cvt_f32_ubyteN). 1/dir is kept finite (trav_init: +-2^60 for an
// axis-parallel ray), so no inf - inf arises [with 1/dir = +-inf the fused form is NaN on every plane of such a ray, NaNs
// are dropped by fminf/fmaxf, and the ray visits the whole tree: 5.4 s for a handful of zero-length directions,
// tests/test_closest_hit_gpu.py::test_degenerate_directions]; a remaining NaN (NaN inputs are turned away by trav_init) only
// makes the test more conservative. The far side is widened (RTX_CULL_SLACK) so rounding never culls a box the ray grazes
// (the reference's scene_box.cpp:8-46 accepts tmin <= tmax inclusively as well).
typedef float rtx_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void trav_inner_step(TravState& s, const DevScene& sc, TravStack& st) {
    const uint4* nodes4 = reinterpret_cast<const uint4*>(sc.nodes);
    const size_t node = (size_t)(s.cur & ~RTX_REF_TRANS);
    const uint4 w0 = nodes4[node * 4 + 0];  // ox oy oz | ex ey ez n
    const uint4 w1 = nodes4[node * 4 + 1];  // child[4]
    const uint4 w2 = nodes4[node * 4 + 2];  // qlo x[4] y[4] z[4] | qhi x[4]
    const uint4 w3 = nodes4[node * 4 + 3];  // qhi y[4] z[4] | pad
    const uint32_t ex = w0.w & 0xFFu, ey = (w0.w >> 8) & 0xFFu, ez = (w0.w >> 16) & 0xFFu;
    // distance to plane q of an axis: (o + q * scale - ro) / dir = q * (scale/dir) + (o - ro)/dir, one fused multiply-add per
    // plane (two planes per v_pk_fma_f32); scale is a power of two, so scale/dir is exact, and 1/dir is finite (trav_init)
    // (RTX_TRAV_SLAB_TWO_OPS: the earlier form, (q - (ro - o)/scale) * (scale/dir), a packed subtract and a packed multiply per
    // pair of planes - kept for kernels_legacy.hip, where the other instruction mix moves a register spill in front of a lane
    // re-enable: tools/check_spill_exec.py, DESIGN.md "Toolchain notes")
    const float qdx = s.idx * scale_from_exp(ex), qdy = s.idy * scale_from_exp(ey), qdz = s.idz * scale_from_exp(ez);
#ifdef RTX_TRAV_SLAB_TWO_OPS
    const float rox = (s.r.ox - __uint_as_float(w0.x)) * __uint_as_float((254u - ex) << 23);
    const float roy = (s.r.oy - __uint_as_float(w0.y)) * __uint_as_float((254u - ey) << 23);
    const float roz = (s.r.oz - __uint_as_float(w0.z)) * __uint_as_float((254u - ez) << 23);
#else
    const float rox = (__uint_as_float(w0.x) - s.r.ox) * s.idx;
    const float roy = (__uint_as_float(w0.y) - s.r.oy) * s.idy;
    const float roz = (__uint_as_float(w0.z) - s.r.oz) * s.idz;
#endif
    // sign bit of 1/dir (also right for -0 -> -inf): negative direction enters through the high plane
    const bool negx = (__float_as_uint(s.idx) >> 31) != 0u, negy = (__float_as_uint(s.idy) >> 31) != 0u,
               negz = (__float_as_uint(s.idz) >> 31) != 0u;
    const uint32_t nqx = negx ? w2.w : w2.x, fqx = negx ? w2.x : w2.w;
    const uint32_t nqy = negy ? w3.x : w2.y, fqy = negy ? w2.y : w3.x;
    const uint32_t nqz = negz ? w3.y : w2.z, fqz = negz ? w2.z : w3.y;
    const rtx_f2 vdx = {qdx, qdx}, vdy = {qdy, qdy}, vdz = {qdz, qdz};
    const rtx_f2 vrx = {rox, rox}, vry = {roy, roy}, vrz = {roz, roz};
    float ct[4];
    uint32_t cr[4] = {w1.x, w1.y, w1.z, w1.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t sh = 8u * i;
        const rtx_f2 qx = {(float)((nqx >> sh) & 0xFFu), (float)((fqx >> sh) & 0xFFu)};
        const rtx_f2 qy = {(float)((nqy >> sh) & 0xFFu), (float)((fqy >> sh) & 0xFFu)};
        const rtx_f2 qz = {(float)((nqz >> sh) & 0xFFu), (float)((fqz >> sh) & 0xFFu)};
#ifdef RTX_TRAV_SLAB_TWO_OPS
        const rtx_f2 tx = (qx - vrx) * vdx;
        const rtx_f2 ty = (qy - vry) * vdy;
        const rtx_f2 tz = (qz - vrz) * vdz;
#else
        const rtx_f2 tx = __builtin_elementwise_fma(qx, vdx, vrx);
        const rtx_f2 ty = __builtin_elementwise_fma(qy, vdy, vry);
        const rtx_f2 tz = __builtin_elementwise_fma(qz, vdz, vrz);
#endif
        const float tn = fmaxf(fmaxf(tx.x, ty.x), fmaxf(tz.x, 0.0f));
        const float tf = fminf(fminf(tx.y, ty.y), fminf(tz.y, s.best.t));
        const bool hit = (cr[i] != RTX_EMPTY_CHILD) && (tn <= tf * RTX_CULL_SLACK);
        ct[i] = hit ? tn : __builtin_inff();
    }
    // sort the four (distance, child) pairs, nearest first (5 compare-exchanges)
#define RTX_CSWAP(a, b)                                     \
    {                                                       \
        const bool s_ = ct[b] < ct[a];                      \
        const float tt_ = s_ ? ct[b] : ct[a];               \
        const float tb_ = s_ ? ct[a] : ct[b];               \
        const uint32_t rt_ = s_ ? cr[b] : cr[a];            \
        const uint32_t rb_ = s_ ? cr[a] : cr[b];            \
        ct[a] = tt_; ct[b] = tb_; cr[a] = rt_; cr[b] = rb_; \
    }
    RTX_CSWAP(0, 1) RTX_CSWAP(2, 3) RTX_CSWAP(0, 2) RTX_CSWAP(1, 3) RTX_CSWAP(1, 2)
#undef RTX_CSWAP
    // far children go on the stack, farthest first. Misses sort to the end (inf), so hits are a prefix of ct[1..3].
    const int h1 = ct[1] != __builtin_inff(), h2 = ct[2] != __builtin_inff(), h3 = ct[3] != __builtin_inff();
    if (s.sp + 3 <= STACK_LDS) {
        // all in LDS: three unconditional writes; a slot written for a miss is overwritten by the next write or
        // stays above the stack top
        st.lds[s.sp * TRAV_BLOCK] = make_uint2(cr[3], __float_as_uint(ct[3]));
        st.lds[(s.sp + h3) * TRAV_BLOCK] = make_uint2(cr[2], __float_as_uint(ct[2]));
        st.lds[(s.sp + h3 + h2) * TRAV_BLOCK] = make_uint2(cr[1], __float_as_uint(ct[1]));
        s.sp += h1 + h2 + h3;
    } else {
        if (h3) st.push(s.sp, cr[3], ct[3]);
        if (h2) st.push(s.sp, cr[2], ct[2]);
        if (h1) st.push(s.sp, cr[1], ct[1]);
    }
    if (ct[0] != __builtin_inff()) s.cur = cr[0];
    else trav_pop(s, st);
}

// The records of one leaf reference (consecutive 48-byte records). Returns the number tested.
template <class SC>
__device__ __forceinline__ uint32_t trav_leaf_test(TravState& s, const SC& sc, uint32_t ref, int hitSide) {
    const float4* tris4 = reinterpret_cast<const float4*>(sc.tris);
    const uint32_t first = (ref & 0x3FFFFFFFu) >> 3;   // without the leaf flag and RTX_REF_TRANS (shadow tree)
    const uint32_t cnt = (ref & 7u) + 1u;
    for (uint32_t k = 0; k < cnt; ++k) {
        const uint32_t slot = first + k;
        const float4 a = tris4[(size_t)slot * 3 + 0];
        const float4 b = tris4[(size_t)slot * 3 + 1];
        const float4 c = tris4[(size_t)slot * 3 + 2];
        test_primitive(sc, a, b, c, s.r, hitSide, slot, s.best, s.bestPrim, s.bestWord, s.excl);
    }
    return cnt;
}
// One leaf: test it, then continue with the next subtree on the stack.
template <class SC>
__device__ __forceinline__ uint32_t trav_leaf_step(TravState& s, const SC& sc, TravStack& st, int hitSide) {
    const uint32_t cnt = trav_leaf_test(s, sc, s.cur, hitSide);
    trav_pop(s, st);
    return cnt;
}

// Closest hit along r within (EPS, r.tmax), run to completion by one lane (shadow chains, photons).
// counters (optional): [0] += nodes visited, [1] += triangles tested.
// The end of a traversal in the general kernels (SC::kXF): does the reference reach the hit it ended with? Its box tests
// come first - the root's child box for the whole scene (IntersectNodeAdapter.h:30-32), the object's own box in node
// coordinates for the object (:63-75) - and a hit within rounding of a box's silhouette fails them now and then
// (ref_box_hit). Returns true when s.best stands (or has become a miss: root box); false when the object is culled for
// this ray, in which case the traversal has been started again without it (TravState::excl) and the caller goes on
// stepping. One object per ray: a second cull in one traversal is not followed (two silhouettes within rounding on one
// ray). The lean kernels (world-space triangles only) return true at once: see ref_box_hit.
template <class SC>
__device__ __forceinline__ bool trav_verify(TravState& s, const SC& sc) {
    if (!SC::kXF) return true;
    if (s.best.slot < 0) return true;
    const RayT r = s.r;
    const uint32_t kind = sc.tris[s.best.slot].kind, obj = sc.tris[s.best.slot].obj & TRI_OBJ_MASK;
    if (kind == 0u) {   // world-space triangle hit well inside both boxes: both tests pass (well_inside_box), no divisions
        const float t = s.best.t, px = r.ox + r.dx * t, py = r.oy + r.dy * t, pz = r.oz + r.dz * t;
        if (t < 1e6f && r.tmax > t * 1.001f && well_inside_box(sc.bounds, px, py, pz, t) && well_inside_box(sc.objXf[obj].box, px, py, pz, t)) return true;
    }
    if (!ref_box_hit(sc.bounds, r.ox, r.oy, r.oz, r.dx, r.dy, r.dz, r.tmax)) {
        s.best.t = r.tmax;
        s.best.u = 0.0f;
        s.best.v = 0.0f;
        s.best.slot = -1;
        s.bestPrim = 0xFFFFFFFFu;
        s.bestWord = 0u;
        return true;
    }
    if (kind != 0u && kind != RTX_KIND_XFTRI) return true;   // Sphere / Plane: tested on the way in (test_analytic)
    float p[3] = {r.ox, r.oy, r.oz}, d[3] = {r.dx, r.dy, r.dz};
    if (kind == RTX_KIND_XFTRI) ray_to_object(sc, obj, r, p, d);
    if (ref_box_hit(sc.objXf[obj].box, p[0], p[1], p[2], d[0], d[1], d[2], r.tmax)) return true;
    if (s.excl != RTX_NO_OBJECT) return true;
    trav_init(s, sc, r);
    s.excl = obj;
    return false;
}

template <bool COUNT, class SC>
__device__ __forceinline__ HitT traverse_closest(const SC& sc, const RayT& r, int hitSide, TravStack st,
                                                 uint32_t* nodeCount, uint32_t* triCount) {
    TravState s;
    trav_init(s, sc, r);
    for (;;) {
        while (s.cur != RTX_TRAV_DONE) {
            while (trav_is_inner(s.cur)) {
                if (COUNT) (*nodeCount)++;
                trav_inner_step(s, sc, st);
            }
            if (s.cur != RTX_TRAV_DONE) {
                const uint32_t n = trav_leaf_step(s, sc, st, hitSide);
                if (COUNT) (*triCount) += n;
            }
        }
        if (trav_verify(s, sc)) break;
    }
    return s.best;
}

// PointLight::IntersectRay (Shading.cpp:65-86) for the renderable lights, applied after the scene hit as in
// IntersectSceneWithLights (IntersectNodeAdapter.h:129-145): accept when 0 < t_light < current best.
__device__ __forceinline__ void intersect_lights(const DevScene& sc, const RayT& r, HitT& best) {
#pragma clang fp contract(off)
    for (int i = 0; i < sc.n_lights; ++i) {
        const DevLight& L = sc.lights[i];
        if (L.size <= 0.0f) continue;
        const float ocx = r.ox - L.pos[0], ocy = r.oy - L.pos[1], ocz = r.oz - L.pos[2];
        const float b = ocx * r.dx + ocy * r.dy + ocz * r.dz;
        const float c = (ocx * ocx + ocy * ocy + ocz * ocz) - L.size * L.size;
        const float disc = b * b - c;
        if (disc < 0.0f) continue;
        float t = -b - sqrtf(fmaxf(0.0f, disc));
        if (t < 0.0f) t = -b + sqrtf(fmaxf(0.0f, disc));
        if (t < 0.0f) continue;
        if (t > 0.0f && t < best.t) {
            best.t = t;
            best.u = 0.0f;
            best.v = 0.0f;
            best.slot = -2 - i;
        }
    }
}
