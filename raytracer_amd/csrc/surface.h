// Device-side reconstruction of the HitInfo fields from a compact hit {t,u,v,slot}, texture lookups and bump
// mapping. Follows the tail of TriObj::IntersectTriangle (reference RayTracer/objects.cpp:172-240),
// PointLight::IntersectRay (RayTracer/Shading.cpp:78-84) and ApplyBumpMapping (RayTracer/BumpMapping.h:9-70).
#pragma once
#include "traverse.h"
#include "vec.h"

struct SurfHit {
    f3 p, N, GN, uvw, dpdu, dpdv;
    float t;
    int obj, mtlID, face, light;  // light = index or -1
    uint32_t triWord;             // the triangle record's `obj` word (TRI_OPAQUE, TRI_INTERIOR); 0 for analytic objects and lights
    bool front;
};

// TextureFile::Eval of the absent framework: tile, then bilinear with wrap-around (stand-in choice).
// The four texels of a bilinear tap and their weights. A tap's texel loads are independent of each other; they are issued
// together under ONE format branch (tex_eval_raw_n) - a format test per texel puts a wait for the load just issued between
// any two of them, and the three taps of a bump lookup were twelve memory round trips in a row (the shade kernel waits on
// memory for 63 % of a wave's lifetime: profiles/r03/pmc_summary_spp8_r03b.txt).
struct TexTap {
    size_t i00, i10, i01, i11;
    float fx, fy;
};
__device__ __forceinline__ TexTap tex_tap(int w, int h, f3 uvw) {
    float ux = uvw.x - (float)(int)uvw.x;
    float uy = uvw.y - (float)(int)uvw.y;
    if (ux < 0.0f) ux += 1.0f;
    if (uy < 0.0f) uy += 1.0f;
    const float x = (float)w * ux;
    const float y = (float)h * uy;
    int ix = (int)x;
    int iy = (int)y;
    TexTap t;
    t.fx = x - (float)ix;
    t.fy = y - (float)iy;
    if (ix >= w) ix -= (ix / w) * w;
    int ixp = ix + 1;
    if (ixp >= w) ixp -= w;
    if (iy >= h) iy -= (iy / h) * h;
    int iyp = iy + 1;
    if (iyp >= h) iyp -= h;
    t.i00 = (size_t)iy * w + ix;
    t.i10 = (size_t)iy * w + ixp;
    t.i01 = (size_t)iyp * w + ix;
    t.i11 = (size_t)iyp * w + ixp;
    return t;
}
__device__ __forceinline__ f3 tex_unorm(uint32_t c) {   // RGBA8 texel, r in the low byte
    return mk3(rtx_unorm8((float)(c & 0xFFu)), rtx_unorm8((float)((c >> 8) & 0xFFu)), rtx_unorm8((float)((c >> 16) & 0xFFu)));   // = (float)c.x / 255.0f ... bit for bit
}
// Texel pointers by address space: the map descriptors reach the shade kernel through a table in LDS, so the compiler cannot see
// that `texels` points to global memory and would issue flat loads (which take the LDS and the vector-memory path both).
typedef float rtx_f4n __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const uint32_t* TexelPtr8;
typedef __attribute__((address_space(1))) const rtx_f4n* TexelPtrF;
__device__ __forceinline__ f3 tex_blend(const TexTap& t, f3 c00, f3 c10, f3 c01, f3 c11) {
    return c00 * ((1 - t.fx) * (1 - t.fy)) + c10 * (t.fx * (1 - t.fy)) + c01 * ((1 - t.fx) * t.fy) + c11 * (t.fx * t.fy);
}
// N bilinear taps of one texture: every texel load of the RGB8 form is in flight before the first conversion.
template <int N>
__device__ __forceinline__ void tex_eval_raw_n(const DevTexMap& tm, const f3 (&uvw)[N], f3 (&out)[N]) {
    const int w = tm.width, h = tm.height;
    if (w + h == 0) {
#pragma unroll
        for (int k = 0; k < N; ++k) out[k] = mk3(0.0f);
        return;
    }
    TexTap t[N];
#pragma unroll
    for (int k = 0; k < N; ++k) t[k] = tex_tap(w, h, uvw[k]);
    if (tm.format == RTX_TEX_RGB8) {
        const TexelPtr8 tx = (TexelPtr8)tm.texels;
        uint32_t c[N][4];
#pragma unroll
        for (int k = 0; k < N; ++k) {
            c[k][0] = tx[t[k].i00];
            c[k][1] = tx[t[k].i10];
            c[k][2] = tx[t[k].i01];
            c[k][3] = tx[t[k].i11];
        }
#pragma unroll
        for (int k = 0; k < N; ++k) out[k] = tex_blend(t[k], tex_unorm(c[k][0]), tex_unorm(c[k][1]), tex_unorm(c[k][2]), tex_unorm(c[k][3]));
    } else {
        const TexelPtrF tx = (TexelPtrF)tm.texels;
#pragma unroll
        for (int k = 0; k < N; ++k) {   // (16-byte texels: one tap's loads at a time, or three taps would hold 48 registers)
            const rtx_f4n c00 = tx[t[k].i00], c10 = tx[t[k].i10], c01 = tx[t[k].i01], c11 = tx[t[k].i11];
            out[k] = tex_blend(t[k], mk3(c00.x, c00.y, c00.z), mk3(c10.x, c10.y, c10.z), mk3(c01.x, c01.y, c01.z), mk3(c11.x, c11.y, c11.z));
        }
    }
}
__device__ __forceinline__ f3 tex_eval_raw(const DevTexMap& tm, f3 uvw) {
    const f3 in[1] = {uvw};
    f3 out[1];
    tex_eval_raw_n<1>(tm, in, out);
    return out[0];
}
__device__ __forceinline__ f3 xf_point3(const float* M, f3 p) {
    return mk3(M[0] * p.x + M[1] * p.y + M[2] * p.z + M[3], M[4] * p.x + M[5] * p.y + M[6] * p.z + M[7],
               M[8] * p.x + M[9] * p.y + M[10] * p.z + M[11]);
}
__device__ __forceinline__ f3 xf_vec3(const float* M, f3 p) {
    return mk3(M[0] * p.x + M[1] * p.y + M[2] * p.z, M[4] * p.x + M[5] * p.y + M[6] * p.z,
               M[8] * p.x + M[9] * p.y + M[10] * p.z);
}
__device__ __forceinline__ f3 tex_eval(const DevScene& sc, int tex, f3 uvw) {
#ifdef RTX_ABLATE_TEX
    return mk3(0.5f);
#endif
    const DevTexMap& tm = sc.texmaps[tex];
    return tex_eval_raw(tm, xf_point3(tm.xf, uvw));
}
// TexturedColor::EvalEnvironment stand-in: rotate by the map transform, lat-long with +z as the pole.
__device__ __forceinline__ f3 tex_eval_env(const DevScene& sc, int tex, f3 dir) {
    const DevTexMap& tm = sc.texmaps[tex];
    const f3 d = safe_normalize3(xf_vec3(tm.xf, dir));
    const float u = rtx_atan2f(d.y, d.x) * (0.5f / 3.14159265358979323846f) + 0.5f;
    const float v = rtx_acosf(fminf(1.0f, fmaxf(-1.0f, d.z))) * (1.0f / 3.14159265358979323846f);
    return tex_eval_raw(tm, mk3(u, v, 0.0f));
}
// TexturedColor::Eval: value * texture (or value alone)
__device__ __forceinline__ f3 eval_tc(const DevScene& sc, const float* value, int tex, f3 uvw) {
    const f3 c = ld3(value);
    if (tex >= 0) return c * tex_eval(sc, tex, uvw);
    return c;
}

// TexturedFloat::Eval stand-in (the framework's texture.h is absent): value * red channel of the texel
__device__ __forceinline__ float eval_tf(const DevScene& sc, float value, int tex, f3 uvw) {
    if (tex >= 0) return value * tex_eval(sc, tex, uvw).x;
    return value;
}

// HitInfo of a Sphere / Plane hit (objects.cpp:47-85, :100-125) followed by Node::FromNodeCoords
// (IntersectNodeAdapter.h:93; framework convention as in the oracle: p by the node transform, normals by the
// transposed inverse and renormalised, dpdu/dpdv as vectors).
__device__ __forceinline__ void resolve_analytic(const DevScene& sc, uint32_t kind, uint32_t obj, const RayT& r, const HitT& h,
                                                 SurfHit& s) {
    float p[3], d[3];
    ray_to_object(sc, obj, r, p, d);
    const f3 hp = mk3(p[0] + d[0] * h.t, p[1] + d[1] * h.t, p[2] + d[2] * h.t);
    f3 GN, dpdu, dpdv, uvw;
    if (kind == RTX_OBJ_SPHERE) {
        GN = normalized3(hp);
        const f3 n = GN;
        const float PIf = 3.14159265358979323846f;
        const float phi = rtx_atan2f(n.y, n.x);
        const float theta = rtx_acosf(fminf(1.0f, fmaxf(-1.0f, n.z)));
        uvw = mk3((phi + PIf) / (2.0f * PIf), 1.0f - (theta / PIf), 0.0f);
        const float sinTheta = rtx_sinf(theta), cosTheta = rtx_cosf(theta), sinPhi = rtx_sinf(phi), cosPhi = rtx_cosf(phi);
        dpdu = mk3(-sinTheta * sinPhi * 2.0f * PIf, sinTheta * cosPhi * 2.0f * PIf, 0.0f);
        dpdv = mk3(-PIf * cosTheta * cosPhi, -PIf * cosTheta * sinPhi, PIf * sinTheta);
    } else {
        GN = mk3(0.0f, 0.0f, 1.0f);
        uvw = mk3(0.5f * (hp.x + 1.0f), 0.5f * (hp.y + 1.0f), 0.0f);
        dpdu = mk3(2.0f, 0.0f, 0.0f);
        dpdv = mk3(0.0f, 2.0f, 0.0f);
    }
    s.front = kind == RTX_OBJ_SPHERE ? (d[0] * GN.x + d[1] * GN.y + d[2] * GN.z) < 0.0f : d[2] < 0.0f;
    // FromNodeCoords (IntersectNodeAdapter.h:93) applies to non-identity nodes only: under the identity the reference keeps
    // the object-space values as they are (renormalising a unit normal again can move its last bit). Written as selects:
    // the identity matrices give the same p / dpdu / dpdv bit for bit, only the renormalised normal needs the choice.
    const bool xf = (sc.objs[obj].flags & OBJ_TRANSFORMED) != 0u;
    const ObjXform& X = sc.objXf[obj];
    const float* F = X.fwd;
    const float* I = X.inv;
    const f3 GNw = normalized3(mk3(I[0] * GN.x + I[4] * GN.y + I[8] * GN.z, I[1] * GN.x + I[5] * GN.y + I[9] * GN.z,
                                   I[2] * GN.x + I[6] * GN.y + I[10] * GN.z));
    const f3 pw = xf_point3(F, hp), duw = xf_vec3(F, dpdu), dvw = xf_vec3(F, dpdv);
    s.p = xf ? pw : hp;
    s.GN = xf ? GNw : GN;
    s.dpdu = xf ? duw : dpdu;
    s.dpdv = xf ? dvw : dpdv;
    s.N = s.GN;
    s.uvw = uvw;
    s.t = h.t;
    s.obj = (int)obj;
    s.mtlID = 0;
    s.face = 0;
    s.light = -1;
}

// HitInfo of a triangle hit: objects.cpp:172-240
__device__ __forceinline__ void resolve_triangle(const DevScene& sc, const RayT& r, const HitT& h, SurfHit& s) {
#ifdef RTX_RESOLVE_TWO_ARRAYS   // (kernels_legacy.hip: the same words from the TriRec itself, as before round 3 - see there)
    const float4* tris4 = reinterpret_cast<const float4*>(sc.tris);
    const float4 a = tris4[(size_t)h.slot * 3 + 0];
    const float4 b = tris4[(size_t)h.slot * 3 + 1];
    const float4 c = tris4[(size_t)h.slot * 3 + 2];
    const uint32_t prim = __float_as_uint(a.w);
#else
    // one 128-byte line: the attributes and the copy of the record's words (layout.h TriAttr), all requested before the first
    // of them is looked at
    const float4* at4 = reinterpret_cast<const float4*>(sc.attrs + h.slot);
    const float4 q0 = at4[0], q1 = at4[1], q2 = at4[2], q3 = at4[3];
    const float4 b = at4[4], c = at4[5];
    const uint32_t prim = sc.attrs[h.slot].prim;
#endif
    const uint32_t kind = __float_as_uint(b.w);
    s.triWord = 0u;
    if (kind != 0u && kind != RTX_KIND_XFTRI) {
        resolve_analytic(sc, kind, __float_as_uint(c.w) & TRI_OBJ_MASK, r, h, s);
        return;
    }
    s.triWord = __float_as_uint(c.w);
    // the ray in the triangle's own space: world for identity nodes, node coordinates otherwise (the HitInfo is then built
    // in object space as objects.cpp:172-240 does and brought back by Node::FromNodeCoords, IntersectNodeAdapter.h:93)
    RayT rr = r;
    if (kind == RTX_KIND_XFTRI) {
        float p[3], d[3];
        ray_to_object(sc, __float_as_uint(c.w) & TRI_OBJ_MASK, r, p, d);
        rr = RayT{p[0], p[1], p[2], d[0], d[1], d[2], r.tmax};
    }
#ifdef RTX_RESOLVE_TWO_ARRAYS
    const float4* at4 = reinterpret_cast<const float4*>(sc.attrs + h.slot);
    const float4 q0 = at4[0], q1 = at4[1], q2 = at4[2], q3 = at4[3];
#endif
    const uint32_t obj_mtl = __float_as_uint(q3.w);
    const int obj = (int)(obj_mtl & 0xFFFFu);
    const ObjRec o = sc.objs[obj];
    const f3 e1 = mk3(b.x, b.y, b.z), e2 = mk3(c.x, c.y, c.z);
    const float u = h.u, v = h.v;
    const float w = 1.0f - u - v;

    f3 GN = cross3(e1, e2);
    const float gnLen2 = len2_3(GN);
    if (gnLen2 > 0.0f) GN = GN / sqrtf(gnLen2);
    else GN = mk3(0.0f, 0.0f, 1.0f);

    f3 N = GN;
    if (o.flags & OBJ_HAS_NORMALS) {
        const f3 n0 = mk3(q0.x, q0.y, q0.z), n1 = mk3(q0.w, q1.x, q1.y), n2 = mk3(q1.z, q1.w, q2.x);
        N = (n0 * w + n1 * u + n2 * v);
        if (len2_3(N) > 0.0f) N = normalized3(N);
        else N = GN;
    }
    const f3 dir = mk3(rr.dx, rr.dy, rr.dz);
    s.front = dot3(dir, GN) < 0.0f;
    s.t = h.t;
    s.p = mk3(rr.ox, rr.oy, rr.oz) + dir * h.t;
    s.GN = GN;
    s.N = N;
    s.obj = obj;
    s.mtlID = (int)(obj_mtl >> 16);
    s.face = (int)prim - o.face_base;
    s.light = -1;
    s.dpdu = mk3(0.0f);
    s.dpdv = mk3(0.0f);
    if (o.flags & OBJ_HAS_TEXCOORDS) {
        const f3 uv0 = mk3(q2.y, q2.z, 0.0f), uv1 = mk3(q2.w, q3.x, 0.0f), uv2 = mk3(q3.y, q3.z, 0.0f);
        const f3 tex = uv0 * w + uv1 * u + uv2 * v;
        s.uvw = mk3(tex.x, tex.y, 0.0f);
        const f3 duv1 = uv1 - uv0;
        const f3 duv2 = uv2 - uv0;
        const float detUV = duv1.x * duv2.y - duv1.y * duv2.x;
        if (fabsf(detUV) > RTX_EPS) {
            const float invDetUV = 1.0f / detUV;
            s.dpdu = (e1 * duv2.y - e2 * duv1.y) * invDetUV;
            s.dpdv = (e2 * duv1.x - e1 * duv2.x) * invDetUV;
        }
    } else {
        s.uvw = mk3(0.0f);
    }
    if (len2_3(s.dpdu) < RTX_EPS || len2_3(s.dpdv) < RTX_EPS) {
        f3 tangent = (fabsf(GN.x) > 0.5f) ? mk3(-GN.y, GN.x, 0.0f) : mk3(0.0f, -GN.z, GN.y);
        tangent = normalized3(tangent);
        f3 bitangent = cross3(GN, tangent);
        bitangent = normalized3(bitangent);
        s.dpdu = tangent;
        s.dpdv = bitangent;
    }
    if (kind == RTX_KIND_XFTRI) {   // Node::FromNodeCoords: p by the node transform, normals by the transposed inverse and renormalised
        const ObjXform& X = sc.objXf[obj];
        const float* F = X.fwd;
        const float* I = X.inv;
        s.p = xf_point3(F, s.p);
        s.N = normalized3(mk3(I[0] * s.N.x + I[4] * s.N.y + I[8] * s.N.z, I[1] * s.N.x + I[5] * s.N.y + I[9] * s.N.z,
                              I[2] * s.N.x + I[6] * s.N.y + I[10] * s.N.z));
        s.GN = normalized3(mk3(I[0] * s.GN.x + I[4] * s.GN.y + I[8] * s.GN.z, I[1] * s.GN.x + I[5] * s.GN.y + I[9] * s.GN.z,
                               I[2] * s.GN.x + I[6] * s.GN.y + I[10] * s.GN.z));
        s.dpdu = xf_vec3(F, s.dpdu);
        s.dpdv = xf_vec3(F, s.dpdv);
    }
}

// HitInfo of a light hit: Shading.cpp:78-84 on top of HitInfo::Init (uvw = 0.5 stand-in)
__device__ __forceinline__ void resolve_light(const DevLight* lights, const RayT& r, const HitT& h, SurfHit& s) {
    const int li = -2 - h.slot;
    const DevLight& L = lights[li];
    const f3 dir = mk3(r.dx, r.dy, r.dz);
    s.t = h.t;
    s.p = mk3(r.ox, r.oy, r.oz) + dir * h.t;
    s.N = safe_normalize3(s.p - ld3(L.pos));
    s.GN = s.N;
    s.mtlID = 0;
    s.obj = -1;
    s.face = -1;
    s.light = li;
    s.front = dot3(dir, s.N) <= 0.0f;
    s.triWord = 0u;
    s.uvw = mk3(0.5f);
    s.dpdu = mk3(0.0f);
    s.dpdv = mk3(0.0f);
}

__device__ __forceinline__ void basis_from_normal(f3 n, f3& t, f3& b) {
    if (fabsf(n.x) > 0.5f) t = cross3(mk3(0.0f, 1.0f, 0.0f), n);
    else t = cross3(mk3(1.0f, 0.0f, 0.0f), n);
    t = normalized3(t);
    b = cross3(n, t);
    b = normalized3(b);
}

// BumpMapping.h:17-70
__device__ __forceinline__ void apply_bump(const DevScene& sc, const DevMaterial& m, SurfHit& h) {
    if (m.bump_tex < 0) return;
    f3 N = h.N;
    if (len2_3(N) <= 0.0f) return;
    N = normalized3(N);
    f3 T = h.dpdu;
    f3 B = h.dpdv;
    if (len2_3(T) < 1e-10f || len2_3(B) < 1e-10f) {
        basis_from_normal(N, T, B);
    } else {
        T = T - N * dot3(T, N);
        if (len2_3(T) < 1e-10f) basis_from_normal(N, T, B);
        else {
            T = normalized3(T);
            B = B - N * dot3(B, N);
            B = B - T * dot3(B, T);
            if (len2_3(B) < 1e-10f) B = cross3(N, T);
            B = normalized3(B);
        }
    }
    const DevTexMap& tm = sc.texmaps[m.bump_tex];
    float du = 0.001f, dv = 0.001f;
    const int maxDim = max(tm.width, tm.height);
    if (maxDim > 0) du = dv = 1.0f / (float)maxDim;
    float h0, hx, hy;
    {
#ifdef RTX_ABLATE_TEX
        h0 = hx = hy = 0.5f;
#else
        const f3 at[3] = {xf_point3(tm.xf, h.uvw), xf_point3(tm.xf, h.uvw + mk3(du, 0.0f, 0.0f)), xf_point3(tm.xf, h.uvw + mk3(0.0f, dv, 0.0f))};
        f3 c[3];
        tex_eval_raw_n<3>(tm, at, c);   // the twelve texels of the three taps in one round trip
        h0 = luma1(c[0]);
        hx = luma1(c[1]);
        hy = luma1(c[2]);
#endif
    }
    const float invDu = (fabsf(du) > 1e-6f) ? 1.0f / du : 0.0f;
    const float invDv = (fabsf(dv) > 1e-6f) ? 1.0f / dv : 0.0f;
    const float dHdU = (hx - h0) * invDu * m.bump_scale;
    const float dHdV = (hy - h0) * invDv * m.bump_scale;
    f3 nT = mk3(-dHdU, -dHdV, 1.0f);
    if (len2_3(nT) > 0.0f) nT = normalized3(nT);
    const f3 bumped = T * nT.x + B * nT.y + N * nT.z;
    if (len2_3(bumped) > 0.0f) h.N = normalized3(bumped);
}
