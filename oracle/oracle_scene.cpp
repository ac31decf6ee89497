// TEST INFRASTRUCTURE ONLY (see oracle_math.h). Ray casting of the CPU restatement.
#include "oracle_scene.h"
#include <numeric>
#include <cstring>

namespace orc {

// ---------------------------------------------------------------------------------------------------------
// BVH2 build: the reference calls cy::BVHTriMesh::SetMesh(this, 4) (objects.cpp:136); cyBVH is absent, so
// the topology is ours: centroid-median split on the widest axis, leaves of <= 4 faces. Closest-hit results do
// not depend on topology except exact-t ties (SURVEY.md §7 "Hard parts").
// ---------------------------------------------------------------------------------------------------------
int Mesh::build_rec(std::vector<int>& idx, int lo, int hi, const std::vector<V3>& cen, const std::vector<Box>& fb) {
    int id = (int)nodes.size();
    nodes.push_back(Node{});
    Box b, cb;
    for (int i = lo; i < hi; ++i) {
        b.add(fb[idx[i]]);
        cb.add(cen[idx[i]]);
    }
    float bb[6] = {b.pmin.x, b.pmin.y, b.pmin.z, b.pmax.x, b.pmax.y, b.pmax.z};
    std::memcpy(nodes[id].b, bb, sizeof(bb));
    int n = hi - lo;
    if (n <= 4) {
        nodes[id].a = (int)elems.size();
        nodes[id].count = n;
        for (int i = lo; i < hi; ++i) elems.push_back(idx[i]);
        return id;
    }
    V3 ext = cb.pmax - cb.pmin;
    int axis = (ext.x >= ext.y && ext.x >= ext.z) ? 0 : (ext.y >= ext.z ? 1 : 2);
    int mid = (lo + hi) / 2;
    std::nth_element(idx.begin() + lo, idx.begin() + mid, idx.begin() + hi, [&](int a, int c) {
        float ca = cen[a][axis], cc = cen[c][axis];
        return ca < cc || (ca == cc && a < c);
    });
    int c0 = build_rec(idx, lo, mid, cen, fb);
    int c1 = build_rec(idx, mid, hi, cen, fb);
    nodes[id].a = c0;
    nodes[id].count = 0;
    nodes[id].c1 = c1;
    return id;
}

void Mesh::build() {
    bounds = Box();
    for (auto& v : V) bounds.add(v);
    nodes.clear();
    elems.clear();
    if (nf == 0) return;
    std::vector<V3> cen(nf);
    std::vector<Box> fb(nf);
    for (int i = 0; i < nf; ++i) {
        V3 a = V[F[3 * i]], b = V[F[3 * i + 1]], c = V[F[3 * i + 2]];
        fb[i].add(a);
        fb[i].add(b);
        fb[i].add(c);
        cen[i] = (fb[i].pmin + fb[i].pmax) * 0.5f;
    }
    std::vector<int> idx(nf);
    std::iota(idx.begin(), idx.end(), 0);
    build_rec(idx, 0, nf, cen, fb);
}

// objects.cpp:140-243. Deviation: dpdu/dpdv are reset on entry exactly as the reference does, but
// the caller keeps the accepted triangle's values (the reference's brute-force path, objects.cpp:259-262);
// in the reference's BVH path a later rejected triangle zeroes them, which depends on the absent cyBVH order.
bool Mesh::intersect_triangle(const Ray& ray, Hit& hInfo, int hitSide, int faceID) const {
    const int* f = &F[3 * faceID];
    const V3 v0 = V[f[0]];
    const V3 v1 = V[f[1]];
    const V3 v2 = V[f[2]];

    const V3 e1 = v1 - v0;
    const V3 e2 = v2 - v0;

    const V3 pvec = cross(ray.dir, e2);
    const float det = dot(e1, pvec);

    if (hitSide == RTX_HIT_FRONT && det <= EPS) return false;
    if (hitSide == RTX_HIT_BACK && det >= -EPS) return false;
    if (std::fabs(det) < EPS) return false;

    const float invDet = 1.0f / det;
    const V3 tvec = ray.p - v0;

    const float u = dot(tvec, pvec) * invDet;
    if (u < 0.0f || u > 1.0f) return false;

    const V3 qvec = cross(tvec, e1);
    const float v = dot(ray.dir, qvec) * invDet;
    if (v < 0.0f || u + v > 1.0f) return false;

    const float t = dot(e2, qvec) * invDet;
    if (t <= EPS || t > hInfo.z) return false;
    // objects.cpp:166 rejects t >= hInfo.z, so among faces at exactly the same t the first one tested wins: in the
    // reference's brute-force loop that is the smaller face index; its BVH order (cyBVH) is absent. The stated tie rule
    //  is the brute-force one, independent of the order this BVH happens to visit faces in.
    if (t == hInfo.z && !(hInfo.face >= 0 && faceID < hInfo.face)) return false;

    const float w = 1.0f - u - v;

    V3 GN = cross(e1, e2);
    const float gnLen2 = len2(GN);
    if (gnLen2 > 0.0f) GN = GN / std::sqrt(gnLen2);
    else GN = V3(0, 0, 1);

    V3 N = GN;
    if (hasN) {
        const int* fn = &FN[3 * faceID];
        const V3 n0 = VN[fn[0]];
        const V3 n1 = VN[fn[1]];
        const V3 n2 = VN[fn[2]];
        N = (n0 * w + n1 * u + n2 * v);
        if (len2(N) > 0.0f) N = normalized(N);
        else N = GN;
    }

    const bool front = (dot(ray.dir, GN) < 0.0f);
    if ((hitSide == RTX_HIT_FRONT && !front) || (hitSide == RTX_HIT_BACK && front)) return false;

    hInfo.z = t;
    hInfo.p = ray.p + ray.dir * t;
    hInfo.GN = GN;
    hInfo.N = N;
    hInfo.front = front;
    hInfo.face = faceID;
    hInfo.bu = u;
    hInfo.bv = v;
    hInfo.dpdu = V3(0, 0, 0);
    hInfo.dpdv = V3(0, 0, 0);

    if (hasT) {
        const int* ft = &FT[3 * faceID];
        // cy::TriMesh::GetTexCoord(face, bc) with bc = (w,u,v)
        V3 tex = VT[ft[0]] * w + VT[ft[1]] * u + VT[ft[2]] * v;
        hInfo.uvw = V3(tex.x, tex.y, 0.0f);
        const V3 uv0 = VT[ft[0]];
        const V3 uv1 = VT[ft[1]];
        const V3 uv2 = VT[ft[2]];
        const V3 duv1 = uv1 - uv0;
        const V3 duv2 = uv2 - uv0;
        const float detUV = duv1.x * duv2.y - duv1.y * duv2.x;
        if (std::fabs(detUV) > EPS) {
            const float invDetUV = 1.0f / detUV;
            hInfo.dpdu = (e1 * duv2.y - e2 * duv1.y) * invDetUV;
            hInfo.dpdv = (e2 * duv1.x - e1 * duv2.x) * invDetUV;
        }
    } else {
        hInfo.uvw = V3(0, 0, 0);
    }

    if (len2(hInfo.dpdu) < EPS || len2(hInfo.dpdv) < EPS) {
        V3 tangent = (std::fabs(GN.x) > 0.5f) ? V3(-GN.y, GN.x, 0.0f) : V3(0.0f, -GN.z, GN.y);
        tangent = normalized(tangent);
        V3 bitangent = cross(GN, tangent);
        bitangent = normalized(bitangent);
        hInfo.dpdu = tangent;
        hInfo.dpdv = bitangent;
    }

    int matIndex = faceMtl.empty() ? 0 : faceMtl[faceID];
    if (matIndex >= 0) hInfo.mtlID = matIndex;
    return true;
}

// triobj_bvh.cpp:13-34 — recursive, unordered, both children always visited.
bool Mesh::trace_bvh(const Ray& ray, Hit& hInfo, int hitSide, int nodeID, uint64_t* counters) const {
    const Node& nd = nodes[nodeID];
    Box nb;
    nb.pmin = V3(nd.b[0], nd.b[1], nd.b[2]);
    nb.pmax = V3(nd.b[3], nd.b[4], nd.b[5]);
    if (counters) counters[0]++;
    if (!nb.intersect(ray, hInfo.z)) return false;
    bool hit = false;
    if (nd.count > 0) {
        for (int i = 0; i < nd.count; ++i) {
            if (counters) counters[1]++;
            hit |= intersect_triangle(ray, hInfo, hitSide, elems[nd.a + i]);
        }
    } else {
        hit |= trace_bvh(ray, hInfo, hitSide, nd.a, counters);
        hit |= trace_bvh(ray, hInfo, hitSide, nd.c1, counters);
    }
    return hit;
}

// objects.cpp:245-266
// objects.cpp:21-86 (unit sphere at the origin of object space)
bool Mesh::intersect_sphere(const Ray& ray, Hit& hInfo, int hitSide) const {
    const V3& p = ray.p;
    const V3& d = ray.dir;
    const float A = dot(d, d);
    const float B = 2.0f * dot(p, d);
    const float C = dot(p, p) - 1.0f;
    const float disc = B * B - 4 * A * C;
    if (disc < 0.0f) return false;
    const float s = std::sqrt(disc);
    const float inv2A = 0.5f / A;
    const float t0 = (-B - s) * inv2A;
    const float t1 = (-B + s) * inv2A;
    float t = -1.0f;
    if (t0 > EPS) t = t0;
    else if (t1 > EPS) t = t1;
    if (t < 0.0f || t >= hInfo.z) return false;
    const V3 hp = p + d * t;
    const V3 GN = normalized(hp);
    const bool front = dot(d, GN) < 0.0f;
    if ((hitSide == RTX_HIT_FRONT && !front) || (hitSide == RTX_HIT_BACK && front)) return false;
    hInfo.z = t;
    hInfo.p = hp;
    hInfo.GN = GN;
    hInfo.N = GN;
    hInfo.front = front;
    const V3 n = normalized(hp);
    const float PIf = 3.14159265358979323846f;
    const float phi = rtx_atan2f(n.y, n.x);
    const float theta = rtx_acosf(std::min(1.0f, std::max(-1.0f, n.z)));
    hInfo.uvw = V3((phi + PIf) / (2.0f * PIf), 1.0f - (theta / PIf), 0.0f);
    const float sinTheta = rtx_sinf(theta), cosTheta = rtx_cosf(theta), sinPhi = rtx_sinf(phi), cosPhi = rtx_cosf(phi);
    hInfo.dpdu = V3(-sinTheta * sinPhi * 2.0f * PIf, sinTheta * cosPhi * 2.0f * PIf, 0.0f);
    hInfo.dpdv = V3(-PIf * cosTheta * cosPhi, -PIf * cosTheta * sinPhi, PIf * sinTheta);
    hInfo.face = 0;
    hInfo.mtlID = 0;
    return true;
}

// objects.cpp:89-126 (square [-1,1]^2 at z = 0, normal +z)
bool Mesh::intersect_plane(const Ray& ray, Hit& hInfo, int hitSide) const {
    const float denom = ray.dir.z;
    if (std::fabs(denom) < EPS) return false;
    const float t = -(ray.p.z) / denom;
    if (t <= EPS || t >= hInfo.z) return false;
    const bool front = denom < 0.0f;
    if ((hitSide == RTX_HIT_FRONT && !front) || (hitSide == RTX_HIT_BACK && front)) return false;
    const V3 hp = ray.p + ray.dir * t;
    if (hp.x < -1.0f || hp.x > 1.0f) return false;
    if (hp.y < -1.0f || hp.y > 1.0f) return false;
    hInfo.z = t;
    hInfo.p = hp;
    hInfo.GN = V3(0, 0, 1);
    hInfo.N = V3(0, 0, 1);
    hInfo.front = front;
    hInfo.uvw = V3(0.5f * (hp.x + 1.0f), 0.5f * (hp.y + 1.0f), 0.0f);
    hInfo.dpdu = V3(2.0f, 0.0f, 0.0f);
    hInfo.dpdv = V3(0.0f, 2.0f, 0.0f);
    hInfo.face = 0;
    hInfo.mtlID = 0;
    return true;
}

bool Mesh::intersect_ray(const Ray& ray, Hit& hInfo, int hitSide, bool useBVH, uint64_t* counters) const {
    if (kind == RTX_OBJ_SPHERE) return intersect_sphere(ray, hInfo, hitSide);
    if (kind == RTX_OBJ_PLANE) return intersect_plane(ray, hInfo, hitSide);
    if (useBVH) {
        if (nf == 0) return false;
        if (!bounds.empty() && !bounds.intersect(ray, hInfo.z)) return false;
        return trace_bvh(ray, hInfo, hitSide, 0, counters);
    }
    bool anyHit = false;
    for (int i = 0; i < nf; ++i) {
        Hit tmp = hInfo;
        if (intersect_triangle(ray, tmp, hitSide, i)) {
            hInfo = tmp;
            anyHit = true;
        }
    }
    return anyHit;
}

// Shading.cpp:65-86
bool Light::intersect(const Ray& ray, Hit& hInfo) const {
    if (size <= 0.0f) return false;
    V3 oc = ray.p - position;
    float b = dot(oc, ray.dir);
    float c = len2(oc) - size * size;
    float disc = b * b - c;
    if (disc < 0.0f) return false;
    float t = -b - std::sqrt(std::max(0.0f, disc));
    if (t < 0.0f) t = -b + std::sqrt(std::max(0.0f, disc));
    if (t < 0.0f) return false;
    hInfo.z = t;
    hInfo.p = ray.p + ray.dir * t;
    hInfo.N = safe_normalize(hInfo.p - position);
    hInfo.GN = hInfo.N;
    hInfo.mtlID = 0;
    hInfo.light = true;
    hInfo.front = (dot(ray.dir, hInfo.N) <= 0.0f);
    return true;
}

// ---- the scene graph as the reference walks it (IntersectNodeAdapter.h:23-97), for scenes deeper than one level ----
// Node::ToNodeCoords / FromNodeCoords and Node::ComputeChildBoundBox belong to the absent framework (scene.h); the
// conventions are the ones the depth-1 walk below already uses: ray p' = itm * p (translation included), dir' = itm * dir
// not renormalised; hit p = tm * p, normals by itm^T and renormalised, dpdu / dpdv by tm; a node's child box is the union
// over its children of the 8 transformed corners of (child's object box + child's child box).
void Scene::set_tree(int n, const int* parent, const float* tm12, const int* mesh) {
    tree.assign((size_t)n, TreeNode());
    treeRoots.clear();
    for (int i = 0; i < n; ++i) {
        TreeNode& t = tree[i];
        t.parent = parent[i];
        std::memcpy(t.tm.m, tm12 + 12 * (size_t)i, sizeof(float) * 12);
        t.identity = is_identity(t.tm);
        t.itm = t.identity ? t.tm : invert(t.tm);
        t.mesh = mesh[i];
        if (t.parent >= 0) tree[t.parent].children.push_back(i);
        else treeRoots.push_back(i);
    }
    rootBox = Box();
    for (int i = n - 1; i >= 0; --i) {   // parents come before their children
        Box b = tree[i].childBox;
        if (tree[i].mesh >= 0) b.add(meshes[tree[i].mesh].bounds);
        if (b.empty()) continue;
        Box& dst = tree[i].parent >= 0 ? tree[tree[i].parent].childBox : rootBox;
        for (int c = 0; c < 8; ++c) {
            V3 p((c & 1) ? b.pmax.x : b.pmin.x, (c & 2) ? b.pmax.y : b.pmin.y, (c & 4) ? b.pmax.z : b.pmin.z);
            dst.add(tree[i].identity ? p : xf_point(tree[i].tm, p));
        }
    }
}

// IntersectNodeLocal (IntersectNodeAdapter.h:23-97); node = -1 is rootNode (identity, no object)
bool Scene::intersect_node_local(int node, const Ray& localRay, Hit& outHit, int hitSide, bool useBVH, uint64_t* counters) const {
    bool anyHit = false;
    Hit bestLocal;
    bestLocal.init();
    bestLocal.z = outHit.z;   // :28
    const Box& childUnionBox = node < 0 ? rootBox : tree[node].childBox;
    const bool childUnionMiss = !childUnionBox.empty() && !childUnionBox.intersect(localRay, bestLocal.z);   // :30-32
    if (node >= 0 && tree[node].mesh >= 0) {   // :35-51
        const Mesh& m = meshes[tree[node].mesh];
        const bool bboxOK = m.bounds.empty() || m.bounds.intersect(localRay, bestLocal.z);
        if (bboxOK) {
            Hit h;
            h.init();
            h.z = bestLocal.z;
            if (m.intersect_ray(localRay, h, hitSide, useBVH, counters)) {
                if (h.z > 0.0f && h.z < bestLocal.z) {
                    bestLocal = h;
                    bestLocal.obj = tree[node].mesh;
                    anyHit = true;
                }
            }
        }
    }
    if (!childUnionMiss) {   // :54-90
        const std::vector<int>& kids = node < 0 ? treeRoots : tree[node].children;
        for (int c : kids) {
            const TreeNode& child = tree[c];
            Ray childRay = localRay;   // :62 child->ToNodeCoords(localRay)
            if (!child.identity) {
                childRay.p = xf_point(child.itm, localRay.p);
                childRay.dir = xf_vec(child.itm, localRay.dir);
            }
            bool childMaybeHit = false;   // :63-75
            if (!child.childBox.empty() && child.childBox.intersect(childRay, bestLocal.z)) childMaybeHit = true;
            if (child.mesh >= 0) {
                const Box& ob = meshes[child.mesh].bounds;
                if (ob.empty() || ob.intersect(childRay, bestLocal.z)) childMaybeHit = true;
            }
            if (!childMaybeHit) continue;
            Hit childLocal;
            childLocal.init();
            childLocal.z = bestLocal.z;
            if (intersect_node_local(c, childRay, childLocal, hitSide, useBVH, counters)) {
                if (childLocal.z > 0.0f && childLocal.z < bestLocal.z) {   // :84
                    bestLocal = childLocal;
                    anyHit = true;
                }
            }
        }
    }
    if (anyHit) {
        if (node >= 0 && !tree[node].identity) {   // :93 node->FromNodeCoords(bestLocal)
            const TreeNode& t = tree[node];
            bestLocal.p = xf_point(t.tm, bestLocal.p);
            bestLocal.N = normalized(xf_vec_t(t.itm, bestLocal.N));
            bestLocal.GN = normalized(xf_vec_t(t.itm, bestLocal.GN));
            bestLocal.dpdu = xf_vec(t.tm, bestLocal.dpdu);
            bestLocal.dpdv = xf_vec(t.tm, bestLocal.dpdv);
        }
        outHit = bestLocal;
    }
    return anyHit;
}

// IntersectNodeAdapter.h:23-108 applied to the depth-1 tree that CollectSceneNodeObjects flattens
// (SceneAdapter.h:8-18): rootNode (no object, identity) with one child node per mesh. With a node tree set
// (orc_scene_set_tree) the walk is the reference's, level by level, and the meshes' own transforms are not used.
bool Scene::intersect_node(const Ray& inRay, Hit& outHit, int hitSide, bool useBVH, uint64_t* counters) const {
    if (!tree.empty()) {   // IntersectNode :100-108 - rootNode's ToNodeCoords is the identity
        Hit localHit;
        localHit.init();
        localHit.z = outHit.z;
        const bool hit = intersect_node_local(-1, inRay, localHit, hitSide, useBVH, counters);
        if (hit) outHit = localHit;
        return hit;
    }
    // IntersectNode :100-108 — root ToNodeCoords is identity
    Hit bestLocal;
    bestLocal.init();
    bestLocal.z = outHit.z;
    bool anyHit = false;
    // :30-32 child-union cull (RT_USE_BBOX=1, g_rtUseBBox=true)
    const bool childUnionMiss = (!rootBox.empty() && !rootBox.intersect(inRay, bestLocal.z));
    if (!childUnionMiss) {
        for (size_t c = 0; c < meshes.size(); ++c) {
            const Mesh& m = meshes[c];
            // :62 child->ToNodeCoords(localRay): p' = itm*(p-pos), dir' = itm*dir (not renormalised)
            Ray childRay = inRay;
            if (!m.identity) {
                childRay.p = xf_point(m.itm, inRay.p);
                childRay.dir = xf_vec(m.itm, inRay.dir);
            }
            // :63-75: child's own child-union box is empty; object box decides
            bool childMaybeHit = m.bounds.empty() || m.bounds.intersect(childRay, bestLocal.z);
            if (!childMaybeHit) continue;
            Hit childLocal;
            childLocal.init();
            childLocal.z = bestLocal.z;
            // IntersectNodeLocal(child): :35-51
            bool hitChild = false;
            {
                bool bboxOK = m.bounds.empty() || m.bounds.intersect(childRay, childLocal.z);
                if (bboxOK) {
                    Hit h;
                    h.init();
                    h.z = childLocal.z;
                    if (m.intersect_ray(childRay, h, hitSide, useBVH, counters)) {
                        if (h.z > 0.0f && h.z < childLocal.z) {
                            childLocal = h;
                            childLocal.obj = (int)c;
                            hitChild = true;
                        }
                    }
                }
                if (hitChild && !m.identity) {
                    // :93 node->FromNodeCoords: p = tm*p+pos, normals by itm^T and renormalised (framework convention;
                    // the legacy mode's bias rules read N unnormalised, ShadeInfoAdapter.h:60-67)
                    childLocal.p = xf_point(m.tm, childLocal.p);
                    childLocal.N = normalized(xf_vec_t(m.itm, childLocal.N));
                    childLocal.GN = normalized(xf_vec_t(m.itm, childLocal.GN));
                    childLocal.dpdu = xf_vec(m.tm, childLocal.dpdu);
                    childLocal.dpdv = xf_vec(m.tm, childLocal.dpdv);
                }
            }
            if (hitChild) {
                if (childLocal.z > 0.0f && childLocal.z < bestLocal.z) {  // :84 strict '<' keeps the first hit
                    bestLocal = childLocal;
                    anyHit = true;
                }
            }
        }
    }
    if (anyHit) outHit = bestLocal;
    return anyHit;
}

// IntersectNodeAdapter.h:111-151
bool Scene::intersect_with_lights(const Ray& ray, Hit& outHit, int hitSide, bool useBVH, uint64_t* counters) const {
    bool anyHit = false;
    Hit bestHit;
    bestHit.init();
    bestHit.z = outHit.z;
    Hit sceneHit;
    sceneHit.init();
    sceneHit.z = bestHit.z;
    if (intersect_node(ray, sceneHit, hitSide, useBVH, counters)) {
        bestHit = sceneHit;
        bestHit.lightIdx = -1;
        anyHit = true;
    }
    for (size_t i = 0; i < lights.size(); ++i) {
        const Light& L = lights[i];
        if (L.renderable()) {
            Hit lightHit;
            lightHit.init();
            lightHit.z = bestHit.z;
            if (L.intersect(ray, lightHit)) {
                if (lightHit.z > 0.0f && lightHit.z < bestHit.z) {
                    bestHit = lightHit;
                    bestHit.lightIdx = (int)i;
                    bestHit.light = true;
                    bestHit.obj = -1;
                    bestHit.face = -1;
                    anyHit = true;
                }
            }
        }
    }
    if (anyHit) outHit = bestHit;
    return anyHit;
}

// TextureFile::Eval of the absent framework (texture.cpp): tile-clamp then bilinear with wrap-around.
V3 TexMap::eval_raw(V3 uvw) const {
    if (w + h == 0) return V3(0, 0, 0);
    float ux = uvw.x - (float)(int)uvw.x;
    float uy = uvw.y - (float)(int)uvw.y;
    if (ux < 0) ux += 1.0f;
    if (uy < 0) uy += 1.0f;
    float x = (float)w * ux;
    float y = (float)h * uy;
    int ix = (int)x;
    int iy = (int)y;
    float fx = x - (float)ix;
    float fy = y - (float)iy;
    if (ix >= w) ix -= (ix / w) * w;
    int ixp = ix + 1;
    if (ixp >= w) ixp -= w;
    if (iy >= h) iy -= (iy / h) * h;
    int iyp = iy + 1;
    if (iyp >= h) iyp -= h;
    return texel(ix, iy) * ((1 - fx) * (1 - fy)) + texel(ixp, iy) * (fx * (1 - fy)) + texel(ix, iyp) * ((1 - fx) * fy) +
           texel(ixp, iyp) * (fx * fy);
}

// TexturedColor::EvalEnvironment (absent framework). Stand-in: rotate the direction by the map transform,
// lat-long lookup with +z as the pole: u = atan2(y,x)/(2pi)+0.5, v = acos(z)/pi.
V3 TexMap::eval_env(V3 dir) const {
    V3 d = safe_normalize(xf_vec(xf, dir));
    float u = rtx_atan2f(d.y, d.x) * (0.5f / PI_F) + 0.5f;
    float v = rtx_acosf(std::min(1.0f, std::max(-1.0f, d.z))) * (1.0f / PI_F);
    return eval_raw(V3(u, v, 0.0f));
}

Scene* scene_from_desc(const rtx_scene_desc* d) {
    Scene* s = new Scene();
    s->meshes.resize(d->n_meshes);
    int base = 0;
    for (int i = 0; i < d->n_meshes; ++i) {
        const rtx_mesh_desc& md = d->meshes[i];
        Mesh& m = s->meshes[i];
        if (md.kind != RTX_OBJ_MESH) {
            // Sphere::GetBoundBox = [-1,1]^3, Plane::GetBoundBox = [-1,1]^2 x {0}
            m.kind = md.kind;
            m.nf = 1;
            m.bounds.add(V3(-1, -1, md.kind == RTX_OBJ_PLANE ? 0.0f : -1.0f));
            m.bounds.add(V3(1, 1, md.kind == RTX_OBJ_PLANE ? 0.0f : 1.0f));
            std::memcpy(m.tm.m, md.xform, sizeof(float) * 12);
            m.identity = is_identity(m.tm);
            m.itm = m.identity ? m.tm : invert(m.tm);
            m.material = md.material;
            s->faceBase.push_back(base);
            base += 1;
            for (int c = 0; c < 8; ++c) {
                V3 p((c & 1) ? m.bounds.pmax.x : m.bounds.pmin.x, (c & 2) ? m.bounds.pmax.y : m.bounds.pmin.y,
                     (c & 4) ? m.bounds.pmax.z : m.bounds.pmin.z);
                s->rootBox.add(m.identity ? p : xf_point(m.tm, p));
            }
            continue;
        }
        m.V.resize(md.nv);
        for (int k = 0; k < md.nv; ++k) m.V[k] = V3(md.v[3 * k], md.v[3 * k + 1], md.v[3 * k + 2]);
        m.nf = md.nf;
        m.F.assign(md.f, md.f + 3 * (size_t)md.nf);
        if (md.vn && md.nvn > 0) {
            m.hasN = true;
            m.VN.resize(md.nvn);
            for (int k = 0; k < md.nvn; ++k) m.VN[k] = V3(md.vn[3 * k], md.vn[3 * k + 1], md.vn[3 * k + 2]);
            const int* fn = md.fn ? md.fn : md.f;
            m.FN.assign(fn, fn + 3 * (size_t)md.nf);
        }
        if (md.vt && md.nvt > 0) {
            m.hasT = true;
            m.VT.resize(md.nvt);
            for (int k = 0; k < md.nvt; ++k) m.VT[k] = V3(md.vt[3 * k], md.vt[3 * k + 1], md.vt[3 * k + 2]);
            const int* ft = md.ft ? md.ft : md.f;
            m.FT.assign(ft, ft + 3 * (size_t)md.nf);
        }
        if (md.face_mtl) m.faceMtl.assign(md.face_mtl, md.face_mtl + md.nf);
        std::memcpy(m.tm.m, md.xform, sizeof(float) * 12);
        m.identity = is_identity(m.tm);
        m.itm = m.identity ? m.tm : invert(m.tm);
        m.material = md.material;
        m.build();
        s->faceBase.push_back(base);
        base += md.nf;
        // rootNode.GetChildBoundBox(): union of children's boxes in root coordinates
        if (!m.bounds.empty()) {
            if (m.identity) s->rootBox.add(m.bounds);
            else
                for (int c = 0; c < 8; ++c) {
                    V3 p((c & 1) ? m.bounds.pmax.x : m.bounds.pmin.x, (c & 2) ? m.bounds.pmax.y : m.bounds.pmin.y,
                         (c & 4) ? m.bounds.pmax.z : m.bounds.pmin.z);
                    s->rootBox.add(xf_point(m.tm, p));
                }
        }
    }
    s->materials.resize(d->n_materials);
    for (int i = 0; i < d->n_materials; ++i) s->materials[i].d = d->materials[i];
    s->texmaps.resize(d->n_texmaps);
    for (int i = 0; i < d->n_texmaps; ++i) {
        const rtx_texmap_desc& td = d->texmaps[i];
        TexMap& t = s->texmaps[i];
        t.format = td.format;
        t.w = td.width;
        t.h = td.height;
        size_t n = (size_t)td.width * td.height * 3;
        if (td.format == RTX_TEX_RGB8) t.u8.assign((const uint8_t*)td.texels, (const uint8_t*)td.texels + n);
        else t.f32.assign((const float*)td.texels, (const float*)td.texels + n);
        std::memcpy(t.xf.m, td.xform, sizeof(float) * 12);
    }
    s->lights.resize(d->n_lights);
    for (int i = 0; i < d->n_lights; ++i) {
        const rtx_light_desc& ld = d->lights[i];
        s->lights[i].position = V3(ld.position[0], ld.position[1], ld.position[2]);
        s->lights[i].intensity = V3(ld.intensity[0], ld.intensity[1], ld.intensity[2]);
        s->lights[i].size = ld.size;
        s->lights[i].attenuation = ld.attenuation;
    }
    s->envColor = V3(d->environment[0], d->environment[1], d->environment[2]);
    s->envTex = d->environment_tex;
    s->bgColor = V3(d->background[0], d->background[1], d->background[2]);
    s->bgTex = d->background_tex;
    s->bgEmission = d->background_emission;
    return s;
}

}  // namespace orc
