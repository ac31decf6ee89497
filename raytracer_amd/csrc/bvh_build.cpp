// Host-side scene flattening and BVH construction (see bvh_build.h, layout.h).
#include "bvh_build.h"
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <thread>

namespace {

struct AABB {
    float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
    void add(const float* p) {
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], p[a]);
            hi[a] = std::max(hi[a], p[a]);
        }
    }
    void add(const AABB& b) {
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], b.lo[a]);
            hi[a] = std::max(hi[a], b.hi[a]);
        }
    }
    float area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (dx < 0 || dy < 0 || dz < 0) return 0.0f;
        return 2.0f * (dx * dy + dy * dz + dz * dx);
    }
};

struct N2 {
    AABB box;
    int left = -1, right = -1;  // inner
    int first = 0, count = 0;   // leaf when count > 0 (range in idx)
};

struct Builder {
    std::vector<AABB> pbox;
    std::vector<float> cen;  // 3 per prim
    std::vector<int> idx;
    std::vector<N2> nodes;
    std::atomic<int> nextNode{0};
    int maxPar = 4;

    int alloc() { return nextNode.fetch_add(1); }

    void build(int id, int lo, int hi, int parDepth) {
        AABB box, cb;
        for (int i = lo; i < hi; ++i) {
            box.add(pbox[idx[i]]);
            cb.add(&cen[3 * (size_t)idx[i]]);
        }
        nodes[id].box = box;
        const int n = hi - lo;
        if (n <= RTX_MAX_LEAF) {
            nodes[id].first = lo;
            nodes[id].count = n;
            return;
        }
        // binned SAH over the three axes
        const int NB = 16;
        int bestAxis = -1, bestSplit = -1;
        float bestCost = 1e30f;
        for (int a = 0; a < 3; ++a) {
            float c0 = cb.lo[a], c1 = cb.hi[a];
            if (!(c1 > c0)) continue;
            float k = NB / (c1 - c0);
            AABB bb[NB];
            int bc[NB] = {0};
            for (int i = lo; i < hi; ++i) {
                int p = idx[i];
                int b = std::min(NB - 1, std::max(0, (int)((cen[3 * (size_t)p + a] - c0) * k)));
                bb[b].add(pbox[p]);
                bc[b]++;
            }
            float rightArea[NB];
            int rightCnt[NB];
            AABB acc;
            int cnt = 0;
            for (int b = NB - 1; b > 0; --b) {
                acc.add(bb[b]);
                cnt += bc[b];
                rightArea[b] = acc.area();
                rightCnt[b] = cnt;
            }
            acc = AABB();
            cnt = 0;
            for (int b = 0; b < NB - 1; ++b) {
                acc.add(bb[b]);
                cnt += bc[b];
                if (cnt == 0 || rightCnt[b + 1] == 0) continue;
                // leaves hold up to 4 triangles: count cost in units of leaf fetches
                float cost = acc.area() * (float)((cnt + RTX_MAX_LEAF - 1) / RTX_MAX_LEAF) +
                             rightArea[b + 1] * (float)((rightCnt[b + 1] + RTX_MAX_LEAF - 1) / RTX_MAX_LEAF);
                if (cost < bestCost) {
                    bestCost = cost;
                    bestAxis = a;
                    bestSplit = b;
                }
            }
        }
        int mid;
        if (bestAxis >= 0) {
            float c0 = cb.lo[bestAxis], c1 = cb.hi[bestAxis];
            float k = NB / (c1 - c0);
            int a = bestAxis, s = bestSplit;
            auto it = std::partition(idx.begin() + lo, idx.begin() + hi, [&](int p) {
                int b = std::min(NB - 1, std::max(0, (int)((cen[3 * (size_t)p + a] - c0) * k)));
                return b <= s;
            });
            mid = (int)(it - idx.begin());
        } else {
            mid = lo;
        }
        if (mid == lo || mid == hi) {
            // degenerate (coincident centroids): split the index range in the middle, ordered by primitive id
            mid = (lo + hi) / 2;
            std::nth_element(idx.begin() + lo, idx.begin() + mid, idx.begin() + hi);
        }
        int l = alloc(), r = alloc();
        nodes[id].left = l;
        nodes[id].right = r;
        if (parDepth < maxPar && n > 32768) {
            std::thread t([&, l, lo, mid, parDepth]() { build(l, lo, mid, parDepth + 1); });
            build(r, mid, hi, parDepth + 1);
            t.join();
        } else {
            build(l, lo, mid, maxPar);
            build(r, mid, hi, maxPar);
        }
    }
};

struct SrcTri {
    float v[3][3];          // analytic object: v[0] = world box min, v[1] = v[2] = world box max
    int obj, face;
    int kind;
};

inline void xf_point(const float* M, const float* p, float* o) {
    o[0] = M[0] * p[0] + M[1] * p[1] + M[2] * p[2] + M[3];
    o[1] = M[4] * p[0] + M[5] * p[1] + M[6] * p[2] + M[7];
    o[2] = M[8] * p[0] + M[9] * p[1] + M[10] * p[2] + M[11];
}
inline bool is_identity(const float* M) {
    static const float I[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    return std::memcmp(M, I, sizeof(I)) == 0;
}

inline double det33(const float* M) {
    const double a = M[0], b = M[1], c = M[2], d = M[4], e = M[5], f = M[6], g = M[8], h = M[9], i = M[10];
    return a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
}
inline void invert34(const float* M, float* R) {
    double a = M[0], b = M[1], c = M[2], d = M[4], e = M[5], f = M[6], g = M[8], h = M[9], i = M[10];
    double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    double id = 1.0 / det;
    double r[9] = {(e * i - f * h) * id, (c * h - b * i) * id, (b * f - c * e) * id,
                   (f * g - d * i) * id, (a * i - c * g) * id, (c * d - a * f) * id,
                   (d * h - e * g) * id, (b * g - a * h) * id, (a * e - b * d) * id};
    for (int k = 0; k < 3; ++k) {
        R[k * 4 + 0] = (float)r[k * 3 + 0];
        R[k * 4 + 1] = (float)r[k * 3 + 1];
        R[k * 4 + 2] = (float)r[k * 3 + 2];
        R[k * 4 + 3] = (float)(-(r[k * 3 + 0] * M[3] + r[k * 3 + 1] * M[7] + r[k * 3 + 2] * M[11]));
    }
}

// exact position of quantised plane q (the kernels test rays in the node's quantised frame, where planes sit at
// exactly o + q*scale)
inline double decode_q(float o, int q, float scale) { return (double)o + (double)q * (double)scale; }

struct Emitter {
    const Builder& B;
    const std::vector<SrcTri>& src;
    const rtx_scene_desc* desc;
    const std::vector<int>& faceBase;
    HostScene& out;
    int maxDepth = 0;
    std::string err;

    void emit_tri(int p) {
        const SrcTri& t = src[p];
        TriRec r;
        if (t.kind == (int)RTX_KIND_XFTRI) {
            // mesh under a non-identity node transform: the object's own fp32 vertices (the kernels carry the ray to node
            // coordinates, IntersectNodeAdapter.h:62); t.v holds the world-space corners the BVH boxes were built from
            const rtx_mesh_desc& m = desc->meshes[t.obj];
            const float* v0 = m.v + 3 * (size_t)m.f[3 * (size_t)t.face];
            const float* v1 = m.v + 3 * (size_t)m.f[3 * (size_t)t.face + 1];
            const float* v2 = m.v + 3 * (size_t)m.f[3 * (size_t)t.face + 2];
            r.v0x = v0[0]; r.v0y = v0[1]; r.v0z = v0[2];
            r.e1x = v1[0] - v0[0]; r.e1y = v1[1] - v0[1]; r.e1z = v1[2] - v0[2];
            r.e2x = v2[0] - v0[0]; r.e2y = v2[1] - v0[1]; r.e2z = v2[2] - v0[2];
        } else {
            r.v0x = t.v[0][0]; r.v0y = t.v[0][1]; r.v0z = t.v[0][2];
            // e1 = v1 - v0, e2 = v2 - v0 in fp32: the operands the reference forms per test (objects.cpp:146-147)
            r.e1x = t.v[1][0] - t.v[0][0]; r.e1y = t.v[1][1] - t.v[0][1]; r.e1z = t.v[1][2] - t.v[0][2];
            r.e2x = t.v[2][0] - t.v[0][0]; r.e2y = t.v[2][1] - t.v[0][1]; r.e2z = t.v[2][2] - t.v[0][2];
        }
        // opaque for shadow rays: no material, or max(Refraction().GetValue()) <= 0 (RayCaster.cpp:215-229)
        const int objMtl = desc->meshes[t.obj].material;
        bool opaque = true;
        if (objMtl >= 0 && objMtl < desc->n_materials) {
            const float* kt = desc->materials[objMtl].refraction;
            opaque = !(std::max(kt[0], std::max(kt[1], kt[2])) > 0.0f);
        }
        r.kind = (uint32_t)t.kind;
        r.obj = (uint32_t)t.obj | (opaque ? TRI_OPAQUE : 0u);
        r.prim = (uint32_t)(faceBase[t.obj] + t.face);
        out.tris.push_back(r);
        if (t.kind != RTX_OBJ_MESH && t.kind != (int)RTX_KIND_XFTRI) {
            TriAttr a;
            std::memset(&a, 0, sizeof(a));
            a.obj_mtl = (uint32_t)t.obj;
            out.attrs.push_back(a);
            return;
        }

        const rtx_mesh_desc& m = desc->meshes[t.obj];
        TriAttr a;
        std::memset(&a, 0, sizeof(a));
        if (m.vn && m.nvn > 0) {
            // object-space normals: for a transformed mesh the interpolated normal goes to the world with the hit
            // (Node::FromNodeCoords, surface.h); the transform is the identity for every cerealBar object
            const int* fn = (m.fn ? m.fn : m.f) + 3 * (size_t)t.face;
            float* dst[3] = {a.n0, a.n1, a.n2};
            for (int k = 0; k < 3; ++k) {
                const float* n = m.vn + 3 * (size_t)fn[k];
                dst[k][0] = n[0]; dst[k][1] = n[1]; dst[k][2] = n[2];
            }
        }
        if (m.vt && m.nvt > 0) {
            const int* ft = (m.ft ? m.ft : m.f) + 3 * (size_t)t.face;
            float* dst[3] = {a.uv0, a.uv1, a.uv2};
            for (int k = 0; k < 3; ++k) {
                const float* uv = m.vt + 3 * (size_t)ft[k];
                dst[k][0] = uv[0]; dst[k][1] = uv[1];
            }
        }
        int mtl = m.face_mtl ? m.face_mtl[t.face] : 0;
        if (mtl < 0) mtl = 0;
        a.obj_mtl = (uint32_t)t.obj | ((uint32_t)(mtl & 0xFFFF) << 16);
        out.attrs.push_back(a);
    }

#if !RTX_BVH8
    uint32_t emit(int id, int depth) {
        const N2& n = B.nodes[id];
        maxDepth = std::max(maxDepth, depth);
        if (n.count > 0) {
            uint32_t first = (uint32_t)out.tris.size();
            for (int i = 0; i < n.count; ++i) emit_tri(B.idx[n.first + i]);
            return RTX_LEAF_FLAG | (first << 3) | (uint32_t)(n.count - 1);
        }
        // collapse: pull up grandchildren of the child with the largest surface area until 4 children
        int c[4] = {n.left, n.right, -1, -1};
        int nc = 2;
        while (nc < 4) {
            int best = -1;
            float bestA = -1.0f;
            for (int i = 0; i < nc; ++i) {
                const N2& ch = B.nodes[c[i]];
                if (ch.count > 0) continue;
                float a = ch.box.area();
                if (a > bestA) {
                    bestA = a;
                    best = i;
                }
            }
            if (best < 0) break;
            int l = B.nodes[c[best]].left, r = B.nodes[c[best]].right;
            c[best] = l;
            c[nc++] = r;
        }
        uint32_t my = (uint32_t)out.nodes.size();
        out.nodes.push_back(Node4{});
        uint32_t refs[4] = {RTX_EMPTY_CHILD, RTX_EMPTY_CHILD, RTX_EMPTY_CHILD, RTX_EMPTY_CHILD};
        for (int i = 0; i < nc; ++i) refs[i] = emit(c[i], depth + 1);
        Node4 nd;
        std::memset(&nd, 0, sizeof(nd));
        nd.nchild = (uint8_t)nc;
        nd.ox = n.box.lo[0]; nd.oy = n.box.lo[1]; nd.oz = n.box.lo[2];
        const float o[3] = {nd.ox, nd.oy, nd.oz};
        uint8_t* ebytes[3] = {&nd.ex, &nd.ey, &nd.ez};
        int Efloor = 30;
        for (int a = 0; a < 3; ++a) {
            float extent = n.box.hi[a] - n.box.lo[a];
            if (extent > 0.0f) {
                int e;
                std::frexp(extent / 255.0f, &e);
                Efloor = std::max(Efloor, e + 127 - 40);
            }
        }
        for (int a = 0; a < 3; ++a) {
            float extent = n.box.hi[a] - n.box.lo[a];
            int E = 1;
            if (extent > 0.0f) {
                int e;
                std::frexp(extent / 255.0f, &e);  // extent/255 = m * 2^e, m in [0.5,1)
                E = e + 127;
            }
            // flat or tiny axes: keep 2^-E and (ray origin - o) * 2^-E well inside the float range (the kernels work
            // in the quantised frame), so never go below 2^-40 of the node's largest quantisation step
            if (E < Efloor) E = Efloor;
            for (;; ++E) {
                if (E > 253) {  // the kernels also need 2^-E (biased exponent 254-E) as a normal float
                    err = "scene extent too large to quantise";
                    return my;
                }
                uint32_t bits = (uint32_t)E << 23;
                float scale;
                std::memcpy(&scale, &bits, 4);
                bool ok = true;
                for (int i = 0; i < 4; ++i) {
                    if (i >= nc) {
                        nd.qlo[a][i] = 255;  // inverted box: never hit
                        nd.qhi[a][i] = 0;
                        continue;
                    }
                    const AABB& cb = B.nodes[c[i]].box;
                    int ql = (int)std::floor(((double)cb.lo[a] - (double)o[a]) / (double)scale);
                    ql = std::min(255, std::max(0, ql));
                    while (ql > 0 && decode_q(o[a], ql, scale) > cb.lo[a]) --ql;
                    int qh = (int)std::ceil(((double)cb.hi[a] - (double)o[a]) / (double)scale);
                    qh = std::max(0, qh);
                    while (qh <= 255 && decode_q(o[a], qh, scale) < cb.hi[a]) ++qh;
                    if (qh > 255) {
                        ok = false;
                        break;
                    }
                    nd.qlo[a][i] = (uint8_t)ql;
                    nd.qhi[a][i] = (uint8_t)qh;
                }
                if (ok) break;
            }
            *ebytes[a] = (uint8_t)E;
        }
        for (int i = 0; i < 4; ++i) nd.child[i] = refs[i];
        // pad[0] bit i: child i's subtree holds a surface that lets light through (shadow_kernel follows only those once
        // it knows an opaque blocker); pad[1]: any of them
        for (int i = 0; i < nc; ++i) {
            bool trans = false;
            if (refs[i] & RTX_LEAF_FLAG) {
                const uint32_t first = (refs[i] & 0x7FFFFFFFu) >> 3, cnt = (refs[i] & 7u) + 1u;
                for (uint32_t k = 0; k < cnt; ++k) trans = trans || (out.tris[first + k].obj & TRI_OPAQUE) == 0u;
            } else
                trans = out.nodes[refs[i]].pad[1] != 0u;
            if (trans) nd.pad[0] |= 1u << i;
        }
        nd.pad[1] = nd.pad[0] != 0u ? 1u : 0u;
        out.nodes[my] = nd;
        return my;
    }
#else
    // ---- 8-wide form (layout.h Node8) ----
    // Quantised planes of up to 8 child boxes along axis a of a node whose box starts at o: the smallest power-of-two step
    // (not below 2^-40 of the node's largest step, Efloor) with which every child's upper plane fits in 8 bits.
    bool quantise_axis(float o, float extent, int Efloor, const AABB* const cb[8], int a, uint8_t qlo[8], uint8_t qhi[8], uint8_t& Eout) {
        int E = 1;
        if (extent > 0.0f) {
            int e;
            std::frexp(extent / 255.0f, &e);  // extent/255 = m * 2^e, m in [0.5,1)
            E = e + 127;
        }
        if (E < Efloor) E = Efloor;
        for (;; ++E) {
            if (E > 253) {  // the kernels also need 2^-E (biased exponent 254-E) as a normal float
                err = "scene extent too large to quantise";
                return false;
            }
            uint32_t bits = (uint32_t)E << 23;
            float scale;
            std::memcpy(&scale, &bits, 4);
            bool ok = true;
            for (int i = 0; i < 8; ++i) {
                if (!cb[i]) {
                    qlo[i] = 255;  // inverted box (and the slot is not in the node's valid mask)
                    qhi[i] = 0;
                    continue;
                }
                int ql = (int)std::floor(((double)cb[i]->lo[a] - (double)o) / (double)scale);
                ql = std::min(255, std::max(0, ql));
                while (ql > 0 && decode_q(o, ql, scale) > cb[i]->lo[a]) --ql;
                int qh = (int)std::ceil(((double)cb[i]->hi[a] - (double)o) / (double)scale);
                qh = std::max(0, qh);
                while (qh <= 255 && decode_q(o, qh, scale) < cb[i]->hi[a]) ++qh;
                if (qh > 255) {
                    ok = false;
                    break;
                }
                qlo[i] = (uint8_t)ql;
                qhi[i] = (uint8_t)qh;
            }
            if (ok) break;
        }
        Eout = (uint8_t)E;
        return true;
    }

    // Fills node `my` from the binary subtree `id`; returns whether the subtree holds a surface that lets light through.
    bool fill8(uint32_t my, int id, int depth) {
        static_assert(RTX_MAX_LEAF <= RTX_LEAF8_MAX, "Node8 masks hold 3 triangles per leaf child");
        const N2& n = B.nodes[id];
        maxDepth = std::max(maxDepth, depth);
        // collapse: open the inner child with the largest surface area until there are 8 children
        int c[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
        int nc;
        if (n.count > 0) {   // the whole scene is one leaf
            c[0] = id;
            nc = 1;
        } else {
            c[0] = n.left;
            c[1] = n.right;
            nc = 2;
            while (nc < 8) {
                int best = -1;
                float bestA = -1.0f;
                for (int i = 0; i < nc; ++i) {
                    const N2& ch = B.nodes[c[i]];
                    if (ch.count > 0) continue;
                    float a = ch.box.area();
                    if (a > bestA) {
                        bestA = a;
                        best = i;
                    }
                }
                if (best < 0) break;
                int l = B.nodes[c[best]].left, r = B.nodes[c[best]].right;
                c[best] = l;
                c[nc++] = r;
            }
        }
        // slots: child towards corner s of the node (bit a of s set = high side on axis a) goes to slot s, greedily by
        // the projection of (child centre - node centre) on the corner's diagonal; the traversal visits slots in the order
        // of "slot XOR ray octant", which is then roughly front to back
        int slotChild[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
        {
            float pc[3];
            for (int a = 0; a < 3; ++a) pc[a] = 0.5f * (n.box.lo[a] + n.box.hi[a]);
            float cost[8][8];
            for (int i = 0; i < nc; ++i) {
                const AABB& b = B.nodes[c[i]].box;
                float d[3];
                for (int a = 0; a < 3; ++a) d[a] = 0.5f * (b.lo[a] + b.hi[a]) - pc[a];
                for (int sl = 0; sl < 8; ++sl) cost[i][sl] = ((sl & 1) ? d[0] : -d[0]) + ((sl & 2) ? d[1] : -d[1]) + ((sl & 4) ? d[2] : -d[2]);
            }
            bool cUsed[8] = {false}, sUsed[8] = {false};
            for (int k = 0; k < nc; ++k) {
                int bi = -1, bs = -1;
                float bv = -1e30f;
                for (int i = 0; i < nc; ++i) {
                    if (cUsed[i]) continue;
                    for (int sl = 0; sl < 8; ++sl)
                        if (!sUsed[sl] && cost[i][sl] > bv) {
                            bv = cost[i][sl];
                            bi = i;
                            bs = sl;
                        }
                }
                cUsed[bi] = true;
                sUsed[bs] = true;
                slotChild[bs] = c[bi];
            }
        }
        Node8 nd;
        std::memset(&nd, 0, sizeof(nd));
        nd.ox = n.box.lo[0]; nd.oy = n.box.lo[1]; nd.oz = n.box.lo[2];
        uint32_t valid = 0, imask = 0, tvalid = 0, ttrans = 0, tmask = 0;
        nd.triBase = (uint32_t)out.tris.size();
        const AABB* cb[8];
        for (int sl = 0; sl < 8; ++sl) {
            cb[sl] = nullptr;
            if (slotChild[sl] < 0) continue;
            const N2& cn = B.nodes[slotChild[sl]];
            cb[sl] = &cn.box;
            valid |= 1u << sl;
            if (cn.count == 0) {
                imask |= 1u << sl;
                continue;
            }
            for (int j = 0; j < cn.count; ++j) {
                const size_t rec = out.tris.size();
                emit_tri(B.idx[cn.first + j]);
                tvalid |= 1u << (3 * sl + j);
                if ((out.tris[rec].obj & TRI_OPAQUE) == 0u) {
                    ttrans |= 1u << (3 * sl + j);
                    tmask |= 1u << sl;
                }
            }
        }
        int kInner = 0;
        for (int sl = 0; sl < 8; ++sl) kInner += (imask >> sl) & 1u;
        const uint32_t childBase = kInner ? (uint32_t)out.nodes.size() : 0u;
        if (kInner) out.nodes.resize(out.nodes.size() + (size_t)kInner);
        nd.childBase = childBase;
        nd.imask = (uint8_t)imask;
        int Efloor = 30;
        for (int a = 0; a < 3; ++a) {
            float extent = n.box.hi[a] - n.box.lo[a];
            if (extent > 0.0f) {
                int e;
                std::frexp(extent / 255.0f, &e);
                Efloor = std::max(Efloor, e + 127 - 40);
            }
        }
        const float o[3] = {nd.ox, nd.oy, nd.oz};
        uint8_t* ebytes[3] = {&nd.ex, &nd.ey, &nd.ez};
        for (int a = 0; a < 3; ++a)
            if (!quantise_axis(o[a], n.box.hi[a] - n.box.lo[a], Efloor, cb, a, nd.qlo[a], nd.qhi[a], *ebytes[a])) return false;
        int ord = 0;
        for (int sl = 0; sl < 8; ++sl) {
            if (!((imask >> sl) & 1u)) continue;
            if (fill8(childBase + (uint32_t)ord, slotChild[sl], depth + 1)) tmask |= 1u << sl;
            ++ord;
        }
        nd.tvalid = tvalid | (tmask << 24);
        nd.ttrans = ttrans | (valid << 24);
        out.nodes[my] = nd;
        return tmask != 0u;
    }
#endif
};

}  // namespace

// The words of each TriRec the shading reads, next to the triangle's attributes (layout.h TriAttr): once the records are final
static void copy_records_to_attrs(HostScene& out) {
    for (size_t i = 0; i < out.tris.size() && i < out.attrs.size(); ++i) {
        const TriRec& r = out.tris[i];
        TriAttr& a = out.attrs[i];
        a.e1x = r.e1x; a.e1y = r.e1y; a.e1z = r.e1z;
        a.kind = r.kind;
        a.e2x = r.e2x; a.e2y = r.e2y; a.e2z = r.e2z;
        a.obj = r.obj;
        a.prim = r.prim;
    }
}

bool build_host_scene(const rtx_scene_desc* desc, HostScene& out, std::string& err) {
    out = HostScene();
    if (!desc || desc->n_meshes < 0) {
        err = "null scene description";
        return false;
    }
    if (desc->n_meshes > 65535) {
        err = "more than 65535 objects";
        return false;
    }
    std::vector<int> faceBase(desc->n_meshes, 0);
    size_t total = 0;
    for (int m = 0; m < desc->n_meshes; ++m) {
        const rtx_mesh_desc& md = desc->meshes[m];
        if (md.kind != RTX_OBJ_MESH && md.kind != RTX_OBJ_SPHERE && md.kind != RTX_OBJ_PLANE) {
            err = "object " + std::to_string(m) + ": unknown kind";
            return false;
        }
        if (md.kind == RTX_OBJ_MESH && (md.nf < 0 || md.nv < 0 || (md.nf > 0 && (!md.v || !md.f)))) {
            err = "mesh " + std::to_string(m) + ": missing vertex/face arrays";
            return false;
        }
        for (int k = 0; k < 12; ++k)
            if (!std::isfinite(md.xform[k])) {
                err = "object " + std::to_string(m) + ": non-finite transform";
                return false;
            }
        if (md.material >= desc->n_materials) {
            err = "mesh " + std::to_string(m) + ": material index out of range";
            return false;
        }
        faceBase[m] = (int)total;
        total += md.kind == RTX_OBJ_MESH ? (size_t)md.nf : 1u;   // an analytic object counts as one primitive
    }
    if (total > (size_t)(1u << 27)) {
        err = "more than 2^27 triangles";
        return false;
    }
    std::vector<SrcTri> src(total);
    std::vector<float> itm(12 * (size_t)std::max(1, desc->n_meshes));
    for (int m = 0; m < desc->n_meshes; ++m) {
        if (is_identity(desc->meshes[m].xform)) std::memcpy(&itm[12 * (size_t)m], desc->meshes[m].xform, 48);
        else invert34(desc->meshes[m].xform, &itm[12 * (size_t)m]);
    }
    out.objs.resize(desc->n_meshes);
    out.objXf.resize(std::max(1, desc->n_meshes));
    for (int m = 0; m < desc->n_meshes; ++m) {
        std::memcpy(out.objXf[m].fwd, desc->meshes[m].xform, 48);
        std::memcpy(out.objXf[m].inv, &itm[12 * (size_t)m], 48);
    }
    AABB rootBox;
    for (int m = 0; m < desc->n_meshes; ++m) {
        const rtx_mesh_desc& md = desc->meshes[m];
        const bool ident = is_identity(md.xform);
        ObjRec& o = out.objs[m];
        o.material = md.material;
        o.flags = ((md.vn && md.nvn > 0) ? OBJ_HAS_NORMALS : 0u) | ((md.vt && md.nvt > 0) ? OBJ_HAS_TEXCOORDS : 0u);
        // The reference intersects in object space (IntersectNodeAdapter.h:62,93): a mesh under a non-identity transform
        // keeps its object-space vertices (RTX_KIND_XFTRI records); only its BVH boxes are in world space
        if (md.kind == RTX_OBJ_MESH && !ident) o.flags |= OBJ_TRANSFORMED | OBJ_XFMESH;
        o.face_base = faceBase[m];
        if (md.kind != RTX_OBJ_MESH) {
            // Sphere: GetBoundBox = [-1,1]^3, Plane: [-1,1]^2 x {0}; corners to root coordinates
            o.flags = OBJ_HAS_TEXCOORDS | (ident ? 0u : OBJ_TRANSFORMED);
            o.nf = 1;
            AABB wb;
            for (int c = 0; c < 8; ++c) {
                float p[3] = {(c & 1) ? 1.0f : -1.0f, (c & 2) ? 1.0f : -1.0f, md.kind == RTX_OBJ_PLANE ? 0.0f : ((c & 4) ? 1.0f : -1.0f)};
                float w[3];
                xf_point(md.xform, p, w);
                if (!std::isfinite(w[0]) || !std::isfinite(w[1]) || !std::isfinite(w[2])) {
                    err = "object " + std::to_string(m) + ": non-finite transform";
                    return false;
                }
                wb.add(w);
            }
            rootBox.add(wb);
            const float ob[6] = {-1.0f, -1.0f, md.kind == RTX_OBJ_PLANE ? 0.0f : -1.0f, 1.0f, 1.0f, md.kind == RTX_OBJ_PLANE ? 0.0f : 1.0f};
            std::memcpy(out.objXf[m].box, ob, sizeof(ob));
            SrcTri& t = src[(size_t)faceBase[m]];
            t.obj = m;
            t.face = 0;
            t.kind = md.kind;
            std::memcpy(t.v[0], wb.lo, 12);
            std::memcpy(t.v[1], wb.hi, 12);
            std::memcpy(t.v[2], wb.hi, 12);
            continue;
        }
        o.nf = md.nf;
        for (int k = 0; k < 3 * md.nv; ++k)
            if (!std::isfinite(md.v[k])) {
                err = "mesh " + std::to_string(m) + ": non-finite vertex";
                return false;
            }
        // object box in local coordinates, then its 8 corners in root coordinates (Node::GetChildBoundBox)
        AABB local;
        for (int k = 0; k < md.nv; ++k) local.add(md.v + 3 * (size_t)k);
        std::memcpy(out.objXf[m].box, local.lo, 12);       // TriObj's box: every vertex (cyTriMesh::ComputeBoundingBox)
        std::memcpy(out.objXf[m].box + 3, local.hi, 12);
        if (md.nv > 0) {
            if (ident) rootBox.add(local);
            else
                for (int c = 0; c < 8; ++c) {
                    float p[3] = {(c & 1) ? local.hi[0] : local.lo[0], (c & 2) ? local.hi[1] : local.lo[1],
                                  (c & 4) ? local.hi[2] : local.lo[2]};
                    float w[3];
                    xf_point(md.xform, p, w);
                    rootBox.add(w);
                }
        }
        for (int f = 0; f < md.nf; ++f) {
            SrcTri& t = src[(size_t)faceBase[m] + f];
            t.obj = m;
            t.face = f;
            t.kind = ident ? RTX_OBJ_MESH : (int)RTX_KIND_XFTRI;
            for (int k = 0; k < 3; ++k) {
                int vi = md.f[3 * (size_t)f + k];
                if (vi < 0 || vi >= md.nv) {
                    err = "mesh " + std::to_string(m) + ": vertex index out of range";
                    return false;
                }
                float* dstv = t.v[k];
                if (ident) std::memcpy(dstv, md.v + 3 * (size_t)vi, 12);
                else {
                    xf_point(md.xform, md.v + 3 * (size_t)vi, dstv);
                    if (!std::isfinite(dstv[0]) || !std::isfinite(dstv[1]) || !std::isfinite(dstv[2])) {
                        err = "mesh " + std::to_string(m) + ": vertex overflows under its transform";
                        return false;
                    }
                }
            }
            if (md.vn && md.nvn > 0) {
                const int* fn = (md.fn ? md.fn : md.f) + 3 * (size_t)f;
                for (int k = 0; k < 3; ++k)
                    if (fn[k] < 0 || fn[k] >= md.nvn) {
                        err = "mesh " + std::to_string(m) + ": normal index out of range";
                        return false;
                    }
            }
            if (md.vt && md.nvt > 0) {
                const int* ft = (md.ft ? md.ft : md.f) + 3 * (size_t)f;
                for (int k = 0; k < 3; ++k)
                    if (ft[k] < 0 || ft[k] >= md.nvt) {
                        err = "mesh " + std::to_string(m) + ": texcoord index out of range";
                        return false;
                    }
            }
        }
    }
    std::memcpy(out.bounds, rootBox.lo, 12);
    std::memcpy(out.bounds + 3, rootBox.hi, 12);
    if (total == 0) {
        out.root = RTX_EMPTY_CHILD;
        return true;
    }
    Builder B;
    B.pbox.resize(total);
    B.cen.resize(3 * total);
    B.idx.resize(total);
    // conditioning of every transform: how far a rounding error of the object-space test can sit from the world-space box
    std::vector<float> kappa(std::max(1, desc->n_meshes), 1.0f);
    for (int m = 0; m < desc->n_meshes; ++m) {
        const float* F = out.objXf[m].fwd;
        const float* I = out.objXf[m].inv;
        float nf = 0.0f, ni = 0.0f;
        for (int r = 0; r < 3; ++r) {
            nf = std::max(nf, std::fabs(F[4 * r]) + std::fabs(F[4 * r + 1]) + std::fabs(F[4 * r + 2]));
            ni = std::max(ni, std::fabs(I[4 * r]) + std::fabs(I[4 * r + 1]) + std::fabs(I[4 * r + 2]));
        }
        kappa[m] = std::max(1.0f, nf * ni);
    }
    for (size_t i = 0; i < total; ++i) {
        AABB b;
        for (int k = 0; k < 3; ++k) b.add(src[i].v[k]);
        if (src[i].kind == (int)RTX_KIND_XFTRI) {
            // the triangle is tested in object space, its box lives in world space: the two agree up to the rounding of
            // the transform, so the box gets a margin of a few ulp of its coordinates, scaled by the transform's conditioning
            float mag = 0.0f;
            for (int a = 0; a < 3; ++a) mag = std::max(mag, std::max(std::fabs(b.lo[a]), std::fabs(b.hi[a])));
            const float pad = 4e-6f * kappa[src[i].obj] * mag + 1e-30f;
            for (int a = 0; a < 3; ++a) {
                b.lo[a] -= pad;
                b.hi[a] += pad;
            }
        }
        B.pbox[i] = b;
        for (int a = 0; a < 3; ++a) B.cen[3 * i + a] = 0.5f * (b.lo[a] + b.hi[a]);
        B.idx[i] = (int)i;
    }
    B.nodes.resize(2 * total + 1);
    int root = B.alloc();
    unsigned hc = std::thread::hardware_concurrency();
    B.maxPar = hc >= 16 ? 5 : (hc >= 8 ? 4 : (hc >= 4 ? 3 : 2));
    B.build(root, 0, (int)total, 0);

    out.tris.reserve(total);
    out.attrs.reserve(total);
    out.nodes.reserve(total / 2 + 16);
    Emitter E{B, src, desc, faceBase, out};
#if RTX_BVH8
    out.nodes.resize(1);
    E.fill8(0u, root, 0);
    out.root = 0u;   // always a node (a scene of one or two triangles: a root with one leaf child)
    out.depth = E.maxDepth;
    if (!E.err.empty()) {
        err = E.err;
        return false;
    }
    copy_records_to_attrs(out);
    return true;
#else
    out.root = E.emit(root, 0);
    out.depth = E.maxDepth;
    if (!E.err.empty()) {
        err = E.err;
        return false;
    }
    // the references of children whose subtree holds a surface that lets light through carry RTX_REF_TRANS (bit 30: free
    // below 2^27 triangles and nodes), so that the traversal's stack entries carry it too (traverse.h)
    for (Node4& n : out.nodes)
        for (int i = 0; i < 4; ++i)
            if (n.child[i] != RTX_EMPTY_CHILD && ((n.pad[0] >> i) & 1u)) n.child[i] |= 0x40000000u;
    // ... and the root reference itself (a shadow hop that starts with a known blocker - the blocker cache - looks at nothing
    // but such subtrees from its first step on)
    if (out.root != RTX_EMPTY_CHILD) {
        bool trans = false;
        if (out.root & RTX_LEAF_FLAG) {
            const uint32_t first = (out.root & 0x7FFFFFFFu) >> 3, cnt = (out.root & 7u) + 1;
            for (uint32_t k = 0; k < cnt; ++k) trans = trans || (out.tris[first + k].obj & TRI_OPAQUE) == 0u;
        } else
            trans = out.nodes[out.root].pad[1] != 0u;
        if (trans) out.root |= 0x40000000u;
    }
    // TRI_INTERIOR (layout.h): world-space triangles that no hit below TRI_INTERIOR_TMAX can place within well_inside_box's margin
    // of a face of the root's box or of their object's box - twice that margin, so that the rounding of a hit point o + d t
    // (a few ulps of its terms) cannot matter
    for (TriRec& r : out.tris) {
        if (r.kind != 0u) continue;
        const uint32_t obj = r.obj & TRI_OBJ_MASK;
        if (obj >= out.objXf.size()) continue;
        const float v[3][3] = {{r.v0x, r.v0y, r.v0z}, {r.v0x + r.e1x, r.v0y + r.e1y, r.v0z + r.e1z}, {r.v0x + r.e2x, r.v0y + r.e2y, r.v0z + r.e2z}};
        bool interior = true;
        const float* boxes[2] = {out.bounds, out.objXf[obj].box};
        for (int b = 0; b < 2 && interior; ++b)
            for (int a = 0; a < 3 && interior; ++a) {
                const float lo = std::min(v[0][a], std::min(v[1][a], v[2][a])), hi = std::max(v[0][a], std::max(v[1][a], v[2][a]));
                const float mag = std::max(std::fabs(lo), std::fabs(hi));
                const float ext = boxes[b][3 + a] - boxes[b][a];
                const float eps = 2.0f * (1e-3f * ext + 1e-5f * (mag + TRI_INTERIOR_TMAX)) + 1e-4f * (1.0f + mag);
                interior = ext > 0.0f && lo > boxes[b][a] + eps && hi < boxes[b][3 + a] - eps;
            }
        if (interior) r.obj |= TRI_INTERIOR;
    }
    copy_records_to_attrs(out);
    return true;
#endif
}

namespace {
struct Validator {
    const HostScene& hs;
    std::vector<uint8_t> seen;
    std::string err;
    // exact box of triangle record i, marked as seen
    AABB tri_box(uint32_t i) {
        AABB b;
        if (i >= hs.tris.size()) {
            err = "leaf range out of bounds";
            return b;
        }
        if (seen[i]++) err = "triangle referenced twice";
        const TriRec& t = hs.tris[i];
        float v0[3] = {t.v0x, t.v0y, t.v0z};
        float v1[3] = {t.v0x + t.e1x, t.v0y + t.e1y, t.v0z + t.e1z};
        float v2[3] = {t.v0x + t.e2x, t.v0y + t.e2y, t.v0z + t.e2z};
        if (t.kind == RTX_KIND_XFTRI) {   // object-space record: its box is the world-space one
            const float* F = hs.objXf[t.obj & TRI_OBJ_MASK].fwd;
            float w0[3], w1[3], w2[3];
            xf_point(F, v0, w0);
            xf_point(F, v1, w1);
            xf_point(F, v2, w2);
            b.add(w0);
            b.add(w1);
            b.add(w2);
        } else {
            b.add(v0);
            b.add(v1);
            b.add(v2);
        }
        return b;
    }
    // is the exact box cb inside the quantised box (o, e, qlo, qhi) of a child? (v1/v2 are re-derived as v0+e, one rounding each)
    bool contains(const float o[3], const uint8_t e[3], const uint8_t qlo[3], const uint8_t qhi[3], const AABB& cb) {
        for (int a = 0; a < 3; ++a) {
            uint32_t bits = (uint32_t)e[a] << 23;
            float scale;
            std::memcpy(&scale, &bits, 4);
            double lo = decode_q(o[a], qlo[a], scale), hi = decode_q(o[a], qhi[a], scale);
            float slack = 4e-7f * std::max(std::fabs(cb.lo[a]), std::fabs(cb.hi[a])) + 1e-30f;
            if (lo > cb.lo[a] + slack || hi < cb.hi[a] - slack) return false;
        }
        return true;
    }
#if RTX_BVH8
    // returns the exact box of the subtree of node `ref`; trans: whether it holds a surface that lets light through
    AABB walk(uint32_t ref, bool* transOut = nullptr) {
        AABB b;
        if (ref >= hs.nodes.size()) {
            err = "node index out of bounds";
            return b;
        }
        const Node8& n = hs.nodes[ref];
        const float o[3] = {n.ox, n.oy, n.oz};
        const uint8_t e[3] = {n.ex, n.ey, n.ez};
        const uint32_t valid = n.ttrans >> 24, tmask = n.tvalid >> 24, tvalid = n.tvalid & 0xFFFFFFu, ttrans = n.ttrans & 0xFFFFFFu;
        uint32_t ord = 0, tri = n.triBase;
        bool any = false;
        if (ttrans & ~tvalid) err = "transmissive bit without a triangle";
        for (int sl = 0; sl < 8; ++sl) {
            const bool inner = (n.imask >> sl) & 1u;
            const uint32_t tb = (tvalid >> (3 * sl)) & 7u;
            if (!((valid >> sl) & 1u)) {
                if (inner || tb) err = "child in a slot that is not marked valid";
                continue;
            }
            AABB cb;
            bool trans = false;
            if (inner) {
                if (tb) err = "slot is both inner and leaf";
                cb = walk(n.childBase + ord++, &trans);
            } else {
                if (tb != 1u && tb != 3u && tb != 7u) err = "leaf triangle bits are not a prefix";
                for (int j = 0; j < 3; ++j)
                    if ((tb >> j) & 1u) {
                        if (tri < hs.tris.size() && (hs.tris[tri].obj & TRI_OPAQUE) == 0u) {
                            trans = true;
                            if (!((ttrans >> (3 * sl + j)) & 1u)) err = "transmissive triangle not flagged";
                        } else if ((ttrans >> (3 * sl + j)) & 1u)
                            err = "opaque triangle flagged transmissive";
                        cb.add(tri_box(tri++));
                    }
            }
            if (!err.empty()) return b;
            if (trans != (((tmask >> sl) & 1u) != 0u)) {
                err = "transmissive-subtree mask is wrong";
                return b;
            }
            any = any || trans;
            const uint8_t ql[3] = {n.qlo[0][sl], n.qlo[1][sl], n.qlo[2][sl]}, qh[3] = {n.qhi[0][sl], n.qhi[1][sl], n.qhi[2][sl]};
            if (!contains(o, e, ql, qh, cb)) {
                err = "quantised child box does not contain its subtree";
                return b;
            }
            b.add(cb);
        }
        if (transOut) *transOut = any;
        return b;
    }
#else
    // returns the exact box of the subtree
    AABB walk(uint32_t ref) {
        AABB b;
        ref &= ~0x40000000u;   // RTX_REF_TRANS
        if (ref & RTX_LEAF_FLAG) {
            uint32_t first = (ref & 0x7FFFFFFFu) >> 3, cnt = (ref & 7u) + 1;
            for (uint32_t i = first; i < first + cnt; ++i) {
                b.add(tri_box(i));
                if (!err.empty()) return b;
            }
            return b;
        }
        if (ref >= hs.nodes.size()) {
            err = "node index out of bounds";
            return b;
        }
        const Node4& n = hs.nodes[ref];
        const float o[3] = {n.ox, n.oy, n.oz};
        const uint8_t e[3] = {n.ex, n.ey, n.ez};
        for (int i = 0; i < 4; ++i) {
            if (n.child[i] == RTX_EMPTY_CHILD) continue;
            AABB cb = walk(n.child[i]);
            if (!err.empty()) return b;
            const uint8_t ql[3] = {n.qlo[0][i], n.qlo[1][i], n.qlo[2][i]}, qh[3] = {n.qhi[0][i], n.qhi[1][i], n.qhi[2][i]};
            if (!contains(o, e, ql, qh, cb)) {
                err = "quantised child box does not contain its subtree";
                return b;
            }
            b.add(cb);
        }
        return b;
    }
#endif
};
}  // namespace

std::string validate_host_scene(const HostScene& hs) {
    if (hs.tris.size() != hs.attrs.size()) return "tris/attrs size mismatch";
    if (hs.tris.empty()) return hs.root == RTX_EMPTY_CHILD ? "" : "empty scene with a root";
    Validator v{hs, std::vector<uint8_t>(hs.tris.size(), 0), ""};
    v.walk(hs.root);
    if (!v.err.empty()) return v.err;
    for (size_t i = 0; i < v.seen.size(); ++i)
        if (v.seen[i] != 1) return "triangle not referenced exactly once";
    return "";
}

// ---- a host model of the kernels' traversal order (rtx_debug_host_walk): which nodes a ray visits and which triangles it
// tests, in the order of traverse.h, in plain C++ on decoded boxes. For the CPU tests of the tree and for counting steps when
// a tree layout is changed; no product path calls it.
namespace {
struct WalkRay {
    float o[3], d[3], id[3], tmax;
};
// slab test against the box {lo, hi}; returns entry distance or a negative number for a miss
inline float walk_slab(const WalkRay& r, const float lo[3], const float hi[3], float best) {
    float tn = 0.0f, tf = best;
    for (int a = 0; a < 3; ++a) {
        float t0 = (lo[a] - r.o[a]) * r.id[a], t1 = (hi[a] - r.o[a]) * r.id[a];
        if (r.id[a] < 0.0f || (r.id[a] == 0.0f && std::signbit(r.id[a]))) std::swap(t0, t1);
        if (t0 == t0) tn = std::max(tn, t0);
        if (t1 == t1) tf = std::min(tf, t1);
    }
    return tn <= tf * 1.00001f ? tn : -1.0f;
}
inline void walk_tri(const HostScene& hs, uint32_t slot, const WalkRay& r, float& best, int& bestSlot, uint32_t& bestPrim) {
    const TriRec& T = hs.tris[slot];
    if (T.kind != 0u) return;   // world-space triangles only (the model is for step counts and tree checks)
    const float px = r.d[1] * T.e2z - r.d[2] * T.e2y, py = r.d[2] * T.e2x - r.d[0] * T.e2z, pz = r.d[0] * T.e2y - r.d[1] * T.e2x;
    const float det = T.e1x * px + T.e1y * py + T.e1z * pz;
    if (std::fabs(det) < 1e-7f) return;
    const float inv = 1.0f / det;
    const float tx = r.o[0] - T.v0x, ty = r.o[1] - T.v0y, tz = r.o[2] - T.v0z;
    const float u = (tx * px + ty * py + tz * pz) * inv;
    if (u < 0.0f || u > 1.0f) return;
    const float qx = ty * T.e1z - tz * T.e1y, qy = tz * T.e1x - tx * T.e1z, qz = tx * T.e1y - ty * T.e1x;
    const float v = (r.d[0] * qx + r.d[1] * qy + r.d[2] * qz) * inv;
    if (v < 0.0f || u + v > 1.0f) return;
    const float t = (T.e2x * qx + T.e2y * qy + T.e2z * qz) * inv;
    if (t <= 1e-7f) return;
    if (t > best || (t == best && (bestSlot < 0 || T.prim >= bestPrim))) return;
    best = t;
    bestSlot = (int)slot;
    bestPrim = T.prim;
}
inline float walk_scale(uint8_t e) {
    uint32_t bits = (uint32_t)e << 23;
    float sc;
    std::memcpy(&sc, &bits, 4);
    return sc;
}
}  // namespace

void host_walk_closest(const HostScene& hs, const float* ray8, float* tOut, int* primOut, uint64_t counts[2]) {
    WalkRay r;
    for (int a = 0; a < 3; ++a) {
        r.o[a] = ray8[a];
        r.d[a] = ray8[4 + a];
        r.id[a] = 1.0f / r.d[a];
    }
    r.tmax = ray8[3];
    float best = r.tmax;
    int bestSlot = -1;
    uint32_t bestPrim = 0xFFFFFFFFu;
    uint64_t nNodes = 0, nTris = 0;
    if (hs.root != RTX_EMPTY_CHILD) {
#if RTX_BVH8
        const uint32_t octInv = 7u ^ ((r.id[0] < 0.0f || std::signbit(r.id[0]) ? 1u : 0u) | (r.id[1] < 0.0f || std::signbit(r.id[1]) ? 2u : 0u) |
                                      (r.id[2] < 0.0f || std::signbit(r.id[2]) ? 4u : 0u));
        struct Grp {
            uint32_t base, mask;   // mask: [31:24] hit slots in priority space (priority = slot ^ octInv, highest first), [7:0] imask
        };
        std::vector<Grp> stack;
        Grp G{hs.root, (1u << (24 + (0u ^ octInv))) | 1u};   // the root as a group of one inner child in slot 0
        static const int ideal = std::getenv("BVH8_IDEAL") ? std::atoi(std::getenv("BVH8_IDEAL")) : 0;
        if (ideal) {   // experiment: per-child entries with entry distances, nearest first, culled when popped
            struct Ent { uint32_t node; float t; };
            std::vector<Ent> st2;
            st2.push_back(Ent{hs.root, 0.0f});
            while (!st2.empty()) {
                const Ent e = st2.back();
                st2.pop_back();
                if (!(e.t <= best * 1.00001f)) continue;
                const Node8& n = hs.nodes[e.node];
                ++nNodes;
                const float o[3] = {n.ox, n.oy, n.oz};
                const float sc[3] = {walk_scale(n.ex), walk_scale(n.ey), walk_scale(n.ez)};
                const uint32_t valid = n.ttrans >> 24, tvalid = n.tvalid & 0xFFFFFFu;
                Ent h[8]; int nh = 0; uint32_t ord = 0;
                struct LT { uint32_t sl; float t; } lf[8]; int nl = 0;
                for (int sl = 0; sl < 8; ++sl) {
                    const bool inner = (n.imask >> sl) & 1u;
                    const uint32_t myOrd = ord;
                    if (inner) ++ord;
                    if (!((valid >> sl) & 1u)) continue;
                    float lo[3], hi[3];
                    for (int a = 0; a < 3; ++a) {
                        lo[a] = o[a] + (float)n.qlo[a][sl] * sc[a];
                        hi[a] = o[a] + (float)n.qhi[a][sl] * sc[a];
                    }
                    const float t = walk_slab(r, lo, hi, best);
                    if (t < 0.0f) continue;
                    if (inner) h[nh++] = Ent{n.childBase + myOrd, t};
                    else lf[nl++] = LT{(uint32_t)sl, t};
                }
                if (ideal == 2) std::sort(lf, lf + nl, [](const LT& a, const LT& b) { return a.t < b.t; });
                for (int i = 0; i < nl; ++i) {
                    if (ideal == 2 && !(lf[i].t <= best * 1.00001f)) continue;
                    uint32_t tm = (7u << (3 * lf[i].sl)) & tvalid;
                    while (tm) {
                        const uint32_t b = (uint32_t)__builtin_ctz(tm);
                        tm &= tm - 1u;
                        ++nTris;
                        walk_tri(hs, n.triBase + (uint32_t)__builtin_popcount(tvalid & ((1u << b) - 1u)), r, best, bestSlot, bestPrim);
                    }
                }
                std::stable_sort(h, h + nh, [](const Ent& a, const Ent& b) { return a.t > b.t; });
                for (int i = 0; i < nh; ++i) st2.push_back(h[i]);
            }
        } else
        for (;;) {
            if ((G.mask >> 24) == 0u) {
                if (stack.empty()) break;
                G = stack.back();
                stack.pop_back();
                continue;
            }
            const uint32_t p = 31u - (uint32_t)__builtin_clz(G.mask);
            G.mask &= ~(1u << p);
            const uint32_t slot = (p - 24u) ^ octInv;
            const uint32_t node = G.base + (uint32_t)__builtin_popcount(G.mask & 0xFFu & ((1u << slot) - 1u));
            if ((G.mask >> 24) != 0u) stack.push_back(G);
            const Node8& n = hs.nodes[node];
            ++nNodes;
            const float o[3] = {n.ox, n.oy, n.oz};
            const float sc[3] = {walk_scale(n.ex), walk_scale(n.ey), walk_scale(n.ez)};
            const uint32_t valid = n.ttrans >> 24, tvalid = n.tvalid & 0xFFFFFFu;
            uint32_t hm = 0;
            for (int sl = 0; sl < 8; ++sl) {
                if (!((valid >> sl) & 1u)) continue;
                float lo[3], hi[3];
                for (int a = 0; a < 3; ++a) {
                    lo[a] = o[a] + (float)n.qlo[a][sl] * sc[a];
                    hi[a] = o[a] + (float)n.qhi[a][sl] * sc[a];
                }
                if (walk_slab(r, lo, hi, best) >= 0.0f) hm |= 1u << sl;
            }
            // triangles of the leaf children that were hit, in record order
            uint32_t tm = 0;
            for (int sl = 0; sl < 8; ++sl)
                if (((hm & ~(uint32_t)n.imask) >> sl) & 1u) tm |= 7u << (3 * sl);
            tm &= tvalid;
            while (tm) {
                const uint32_t b = (uint32_t)__builtin_ctz(tm);
                tm &= tm - 1u;
                ++nTris;
                walk_tri(hs, n.triBase + (uint32_t)__builtin_popcount(tvalid & ((1u << b) - 1u)), r, best, bestSlot, bestPrim);
            }
            uint32_t hi8 = hm & n.imask, P = 0;
            for (int sl = 0; sl < 8; ++sl)
                if ((hi8 >> sl) & 1u) P |= 1u << ((uint32_t)sl ^ octInv);
            G = Grp{n.childBase, (P << 24) | n.imask};
        }
#else
        struct Ent {
            uint32_t ref;
            float t;
        };
        std::vector<Ent> stack;
        uint32_t cur = hs.root;
        for (;;) {
            cur &= ~0x40000000u;
            if (cur & RTX_LEAF_FLAG) {
                const uint32_t first = (cur & 0x3FFFFFFFu) >> 3, cnt = (cur & 7u) + 1u;
                for (uint32_t k = 0; k < cnt; ++k) {
                    ++nTris;
                    walk_tri(hs, first + k, r, best, bestSlot, bestPrim);
                }
            } else {
                const Node4& n = hs.nodes[cur];
                ++nNodes;
                const float o[3] = {n.ox, n.oy, n.oz};
                const float sc[3] = {walk_scale(n.ex), walk_scale(n.ey), walk_scale(n.ez)};
                Ent h[4];
                int nh = 0;
                for (int i = 0; i < 4; ++i) {
                    if (n.child[i] == RTX_EMPTY_CHILD) continue;
                    float lo[3], hi[3];
                    for (int a = 0; a < 3; ++a) {
                        lo[a] = o[a] + (float)n.qlo[a][i] * sc[a];
                        hi[a] = o[a] + (float)n.qhi[a][i] * sc[a];
                    }
                    const float t = walk_slab(r, lo, hi, best);
                    if (t >= 0.0f) h[nh++] = Ent{n.child[i], t};
                }
                std::stable_sort(h, h + nh, [](const Ent& a, const Ent& b) { return a.t < b.t; });
                for (int i = nh - 1; i >= 1; --i) stack.push_back(h[i]);
                if (nh > 0) {
                    cur = h[0].ref;
                    continue;
                }
            }
            bool found = false;
            while (!stack.empty()) {
                const Ent e = stack.back();
                stack.pop_back();
                if (e.t <= best * 1.00001f) {
                    cur = e.ref;
                    found = true;
                    break;
                }
            }
            if (!found) break;
        }
#endif
    }
    *tOut = bestSlot >= 0 ? best : -1.0f;
    *primOut = bestSlot >= 0 ? (int)bestPrim : -1;
    counts[0] += nNodes;
    counts[1] += nTris;
}
