// TEST INFRASTRUCTURE ONLY (see oracle_math.h). Scene containers + ray casting of the CPU restatement.
//
// Restates, function by function (citations are paths inside the reference repository):
//   Box::IntersectRay                      RayTracer/scene_box.cpp:8-46
//   TriObj::IntersectTriangle              RayTracer/objects.cpp:140-243
//   TriObj::IntersectRay / brute force     RayTracer/objects.cpp:245-266
//   Sphere::IntersectRay / Plane::IntersectRay  RayTracer/objects.cpp:21-126
//   TriObj::TraceBVHNode                   RayTracer/triobj_bvh.cpp:13-34
//   IntersectNodeLocal / IntersectNode     RayTracer/IntersectNodeAdapter.h:23-108
//   IntersectSceneWithLights               RayTracer/IntersectNodeAdapter.h:111-151
//   PointLight::IntersectRay               RayTracer/Shading.cpp:65-86
// The course framework (cyBVH, cyTriMesh, scene.h ...) is absent from the reference: the BVH
// topology, HitInfo::Init values and Node transform conventions are stand-in choices listed in DESIGN.md.
#pragma once
#include <vector>
#include <memory>
#include "oracle_math.h"
#include "../include/rtx.h"

namespace orc {

constexpr float EPS = 1e-7f;  // objects.cpp:12

struct Ray {
    V3 p, dir;
};

struct Box {
    V3 pmin{BIGFLOAT, BIGFLOAT, BIGFLOAT}, pmax{-BIGFLOAT, -BIGFLOAT, -BIGFLOAT};
    bool empty() const { return pmin.x > pmax.x || pmin.y > pmax.y || pmin.z > pmax.z; }
    void add(V3 p) {
        pmin = V3(std::min(pmin.x, p.x), std::min(pmin.y, p.y), std::min(pmin.z, p.z));
        pmax = V3(std::max(pmax.x, p.x), std::max(pmax.y, p.y), std::max(pmax.z, p.z));
    }
    void add(const Box& b) {
        if (b.empty()) return;
        add(b.pmin);
        add(b.pmax);
    }
    // scene_box.cpp:8-46
    bool intersect(const Ray& r, float t_max) const {
        if (empty()) return false;
        float tmin = 0.0f;
        float tmax = t_max;
        for (int i = 0; i < 3; ++i) {
            const float orig = r.p[i];
            const float dir = r.dir[i];
            const float minv = pmin[i];
            const float maxv = pmax[i];
            if (std::fabs(dir) < 1e-20f) {
                if (orig < minv || orig > maxv) return false;
                continue;
            }
            const float invD = 1.0f / dir;
            float t0 = (minv - orig) * invD;
            float t1 = (maxv - orig) * invD;
            if (t0 > t1) std::swap(t0, t1);
            if (t0 > tmin) tmin = t0;
            if (t1 < tmax) tmax = t1;
            if (tmax < tmin) return false;
        }
        return tmax >= 0.0f && tmin <= t_max;
    }
};

// HitInfo fields read by the hot path + ExtendedHitInfo::hitLight.
struct Hit {
    float z;
    int obj, face, mtlID;
    V3 p, N, GN, uvw, dpdu, dpdv;
    float bu, bv;
    bool front, light;
    int lightIdx;
    void init() {
        z = BIGFLOAT;
        obj = -1;
        face = -1;
        mtlID = 0;
        p = N = GN = dpdu = dpdv = V3(0, 0, 0);
        uvw = V3(0.5f, 0.5f, 0.5f);
        bu = bv = 0;
        front = true;
        light = false;
        lightIdx = -1;
    }
};

struct Mesh {
    std::vector<V3> V, VN, VT;
    std::vector<int> F, FN, FT, faceMtl;  // 3*nf
    int nf = 0;
    bool hasN = false, hasT = false;
    Box bounds;
    M34 tm, itm;
    bool identity = true;
    int material = -1;
    int kind = RTX_OBJ_MESH;   // RTX_OBJ_SPHERE / RTX_OBJ_PLANE: analytic object in object space (nf = 1)
    // own BVH2, leaf <= 4 faces (objects.cpp:136 bvh.SetMesh(this,4))
    struct Node {
        float b[6];
        int a;      // inner: first child ; leaf: first element
        int count;  // 0 = inner
        int c1;     // inner: second child
    };
    std::vector<Node> nodes;
    std::vector<int> elems;

    void build();
    int build_rec(std::vector<int>& idx, int lo, int hi, const std::vector<V3>& cen, const std::vector<Box>& fb);
    bool intersect_triangle(const Ray& ray, Hit& h, int hitSide, int faceID) const;
    bool trace_bvh(const Ray& ray, Hit& h, int hitSide, int nodeID, uint64_t* counters) const;
    bool intersect_ray(const Ray& ray, Hit& h, int hitSide, bool useBVH, uint64_t* counters) const;
    bool intersect_sphere(const Ray& ray, Hit& h, int hitSide) const;   // Sphere::IntersectRay objects.cpp:21-86
    bool intersect_plane(const Ray& ray, Hit& h, int hitSide) const;    // Plane::IntersectRay  objects.cpp:89-126
};

struct TexMap {
    int format, w, h;
    std::vector<uint8_t> u8;
    std::vector<float> f32;
    M34 xf;
    V3 texel(int x, int y) const {
        size_t i = (size_t(y) * w + x) * 3;
        if (format == RTX_TEX_RGB8) return V3(u8[i] / 255.0f, u8[i + 1] / 255.0f, u8[i + 2] / 255.0f);
        return V3(f32[i], f32[i + 1], f32[i + 2]);
    }
    V3 eval_raw(V3 uvw) const;  // TextureFile::Eval: tile + bilinear (stand-in choice)
    V3 eval(V3 uvw) const { return eval_raw(xf_point(xf, uvw)); }
    V3 eval_env(V3 dir) const;  // lat-long (stand-in choice)
};

struct Material {
    rtx_material_desc d;
};

struct Light {
    V3 position, intensity;
    float size, attenuation;
    bool renderable() const { return size > 0.0f; }
    bool intersect(const Ray& ray, Hit& h) const;  // Shading.cpp:65-86
};

// A node of the scene graph below rootNode (scene.h of the absent framework: a transform to the parent, an optional object,
// children). Set by orc_scene_set_tree; without it the scene is the depth-1 tree CollectSceneNodeObjects flattens.
struct TreeNode {
    int parent = -1;              // index of the parent node, -1 = child of rootNode
    M34 tm, itm;                  // node -> parent and its inverse (Node::TransformFrom / TransformTo)
    bool identity = true;
    int mesh = -1;                // Node::GetNodeObj(): index into Scene::meshes (geometry in the node's own coordinates), -1 = none
    std::vector<int> children;
    Box childBox;                 // Node::GetChildBoundBox(): the children's boxes (objects and their child boxes) in this node's coordinates
};

struct Scene {
    std::vector<Mesh> meshes;
    std::vector<Material> materials;
    std::vector<TexMap> texmaps;
    std::vector<Light> lights;
    V3 envColor, bgColor;
    int envTex = -1, bgTex = -1;
    float bgEmission = 0;
    Box rootBox;                  // rootNode.GetChildBoundBox()
    std::vector<int> faceBase;    // global triangle index base per mesh

    // the real node tree (IntersectNodeAdapter.h:23-97 level by level) when orc_scene_set_tree has been called
    std::vector<TreeNode> tree;
    std::vector<int> treeRoots;   // children of rootNode
    void set_tree(int n, const int* parent, const float* tm12, const int* mesh);
    bool intersect_node_local(int node, const Ray& localRay, Hit& out, int hitSide, bool useBVH, uint64_t* counters) const;

    bool intersect_node(const Ray& ray, Hit& out, int hitSide, bool useBVH, uint64_t* counters = nullptr) const;
    bool intersect_with_lights(const Ray& ray, Hit& out, int hitSide, bool useBVH = true,
                               uint64_t* counters = nullptr) const;
    V3 eval_tc(const float value[3], int tex, V3 uvw) const {
        V3 c(value[0], value[1], value[2]);
        if (tex >= 0) return c * texmaps[tex].eval(uvw);
        return c;
    }
    // TexturedFloat::Eval stand-in (texture.h is absent): value * red channel of the texel
    float eval_tf(float value, int tex, V3 uvw) const {
        if (tex >= 0) return value * texmaps[tex].eval(uvw).x;
        return value;
    }
};

Scene* scene_from_desc(const rtx_scene_desc* d);

}  // namespace orc
