// Mirror of the reference's RayTracer/SceneAdapter.h: CollectSceneNodeObjects flattens the node tree (DFS
// pre-order, SceneAdapter.h:8-18). It is the hook where the scene is handed to the GPU: GpuScene walks the same
// list, accumulates the node transforms, and builds the rtx_scene (flattened BVH in HBM) through the C ABI.
#pragma once
#include <map>
#include <stdexcept>
#include "scene.h"

namespace rtamd {

using std::pair;
using std::vector;

// SceneAdapter.h:8-18 (same signature)
inline void CollectSceneNodeObjects(const Scene& scene, vector<pair<const Node*, const Object*>>& nodeObjects) {
    auto collect = [&](const Node* node, auto& collect_ref) -> void {
        if (!node) return;
        if (node->GetNodeObj()) nodeObjects.emplace_back(node, node->GetNodeObj());
        for (int i = 0; i < node->GetNumChild(); ++i) collect_ref(node->GetChild(i), collect_ref);
    };
    collect(&scene.rootNode, collect);
}

// The same DFS with the accumulated node -> root transform of every (Node, Object) pair.
struct SceneNodeItem {
    const Node* node;
    Matrix34 world;
};
inline void CollectSceneNodeItems(const Scene& scene, std::vector<SceneNodeItem>& items, const Node* root = nullptr) {
    auto walk = [&](const Node* n, const Matrix34& parent, auto& self) -> void {
        if (!n) return;
        Matrix34 w = Matrix34::Compose(parent, n->tm);
        if (n->GetNodeObj()) items.push_back({n, w});
        for (int i = 0; i < n->GetNumChild(); ++i) self(n->GetChild(i), w, self);
    };
    walk(root ? root : &scene.rootNode, Matrix34(), walk);
}

// The GPUs this process renders on, set by the host before the first frame like the reference's run-time globals
// (g_rtUseBBox, g_rtUseMeshBVH: Main.cpp:24-26). Empty or one entry: the current HIP device only. Several entries: every
// scene handed to the GPU is replicated on those devices (rtx_group), RayCaster::CastRays deals the 16x16 buckets
// to them as the reference deals them to its worker threads (RayCaster.cpp:1281-1312) and PhotonMapBuilder splits the
// photon ids; single-ray queries (IntersectNode, TraceRay ...) run on the first device of the list, which must be the
// calling thread's current device.
inline std::vector<int> g_rtDevices;

// Device-resident copy of a Scene (or of a sub-tree rooted at `root`). Owns the rtx_scene / rtx_group handle.
class GpuScene {
public:
    GpuScene(const Scene& scene, const Node* root = nullptr) { Build(scene, root ? root : &scene.rootNode); }
    ~GpuScene() {
        if (group) rtx_group_destroy(group);   // owns the member scenes, `handle` among them
        else rtx_scene_destroy(handle);
    }
    GpuScene(const GpuScene&) = delete;
    GpuScene& operator=(const GpuScene&) = delete;
    rtx_scene* Handle() const { return handle; }
    rtx_group* Group() const { return group; }   // non-null when the scene lives on several GPUs (g_rtDevices)
    // (Node, Object) pair of flattened object index i, in CollectSceneNodeObjects order (TriObj, Sphere, Plane)
    const Node* NodeOf(int obj) const { return obj >= 0 && obj < (int)nodes.size() ? nodes[obj] : nullptr; }
    int IndexOf(const Node* n) const {
        for (size_t i = 0; i < nodes.size(); ++i)
            if (nodes[i] == n) return (int)i;
        return -1;
    }
    const std::vector<Light*>& Lights() const { return lights; }

private:
    rtx_scene* handle = nullptr;
    rtx_group* group = nullptr;
    std::vector<const Node*> nodes;
    std::vector<Light*> lights;

    struct TexPool {
        std::vector<rtx_texmap_desc> descs;
        std::map<const TextureMap*, int> index;
        int Get(const TextureMap* t) {
            if (!t) return -1;
            auto it = index.find(t);
            if (it != index.end()) return it->second;
            rtx_texmap_desc d{};
            d.format = t->isFloat ? RTX_TEX_RGB32F : RTX_TEX_RGB8;
            d.width = t->width;
            d.height = t->height;
            d.texels = t->isFloat ? (const void*)t->rgb32f.data() : (const void*)t->rgb8.data();
            for (int k = 0; k < 12; ++k) d.xform[k] = t->itm.m[k];
            descs.push_back(d);
            return index[t] = (int)descs.size() - 1;
        }
    };

    void Build(const Scene& scene, const Node* root) {
        // DFS with accumulated transforms; the (Node,Object) order is CollectSceneNodeObjects' order
        std::vector<SceneNodeItem> items;
        CollectSceneNodeItems(scene, items, root);

        TexPool tex;
        std::map<const Material*, int> mtlIndex;
        std::vector<rtx_material_desc> mtls;
        auto color = [](float* dst, Color const& c) { dst[0] = c.r; dst[1] = c.g; dst[2] = c.b; };
        auto mtlOf = [&](const Material* m) -> int {
            if (!m) return -1;
            auto it = mtlIndex.find(m);
            if (it != mtlIndex.end()) return it->second;
            rtx_material_desc d{};
            d.type = m->Type();
            color(d.diffuse, m->diffuse.value);       d.diffuse_tex = tex.Get(m->diffuse.map.get());
            color(d.specular, m->specular.value);     d.specular_tex = tex.Get(m->specular.map.get());
            color(d.reflection, m->reflection.value); d.reflection_tex = tex.Get(m->reflection.map.get());
            color(d.refraction, m->refraction.value); d.refraction_tex = tex.Get(m->refraction.map.get());
            color(d.emission, m->emission.value);     d.emission_tex = tex.Get(m->emission.map.get());
            d.glossiness = m->glossiness;
            d.glossiness_tex = tex.Get(m->glossinessMap.get());
            d.ior = m->ior;
            color(d.absorption, m->absorption);
            d.bump_tex = tex.Get(m->bump.get());
            d.bump_scale = m->bumpIntensity;
            mtls.push_back(d);
            return mtlIndex[m] = (int)mtls.size() - 1;
        };

        std::vector<rtx_mesh_desc> meshes;
        for (auto const& it : items) {
            auto tri = dynamic_cast<const TriObj*>(it.node->GetNodeObj());
            if (!tri) {
                const bool sphere = dynamic_cast<const Sphere*>(it.node->GetNodeObj()) != nullptr;
                if (!sphere && !dynamic_cast<const Plane*>(it.node->GetNodeObj())) continue;   // unknown Object subclass
                rtx_mesh_desc d{};
                d.kind = sphere ? RTX_OBJ_SPHERE : RTX_OBJ_PLANE;
                for (int k = 0; k < 12; ++k) d.xform[k] = it.world.m[k];
                d.material = mtlOf(it.node->GetMaterial());
                meshes.push_back(d);
                nodes.push_back(it.node);
                continue;
            }
            static_assert(sizeof(Vec3f) == 12 && sizeof(TriObj::TriFace) == 12, "tightly packed arrays expected");
            rtx_mesh_desc d{};
            d.v = &tri->v.data()->x;  d.nv = (int)tri->v.size();
            d.vn = tri->vn.empty() ? nullptr : &tri->vn.data()->x;  d.nvn = (int)tri->vn.size();
            d.vt = tri->vt.empty() ? nullptr : &tri->vt.data()->x;  d.nvt = (int)tri->vt.size();
            d.f = tri->f.empty() ? nullptr : tri->f.data()->v;
            d.fn = tri->fn.empty() ? nullptr : tri->fn.data()->v;
            d.ft = tri->ft.empty() ? nullptr : tri->ft.data()->v;
            d.face_mtl = tri->faceMtl.empty() ? nullptr : tri->faceMtl.data();
            d.nf = (int)tri->f.size();
            for (int k = 0; k < 12; ++k) d.xform[k] = it.world.m[k];
            d.material = mtlOf(it.node->GetMaterial());
            meshes.push_back(d);
            nodes.push_back(it.node);
        }
        std::vector<rtx_light_desc> ls;
        for (Light* l : scene.lights) {
            auto pl = dynamic_cast<PointLight*>(l);
            if (!pl) continue;    // the reference's hot path implements PointLight only (Shading.cpp:64-209)
            rtx_light_desc d{};
            d.position[0] = pl->position.x; d.position[1] = pl->position.y; d.position[2] = pl->position.z;
            color(d.intensity, pl->intensity);
            d.size = pl->size;
            d.attenuation = pl->attenuation;
            ls.push_back(d);
            lights.push_back(l);
        }
        rtx_scene_desc sd{};
        color(sd.environment, scene.environment.value);
        sd.environment_tex = tex.Get(scene.environment.map.get());
        color(sd.background, scene.background.value);
        sd.background_tex = tex.Get(scene.background.map.get());
        sd.background_emission = scene.backgroundEmission;
        sd.meshes = meshes.data();      sd.n_meshes = (int)meshes.size();
        sd.materials = mtls.data();     sd.n_materials = (int)mtls.size();
        sd.texmaps = tex.descs.data();  sd.n_texmaps = (int)tex.descs.size();
        sd.lights = ls.data();          sd.n_lights = (int)ls.size();
        if (g_rtDevices.size() > 1) {
            if (rtx_group_create(&sd, g_rtDevices.data(), (int)g_rtDevices.size(), &group) != RTX_OK)
                throw std::runtime_error(std::string("rtx_group_create: ") + rtx_last_error());
            handle = rtx_group_scene(group, 0);
        } else if (rtx_scene_create(&sd, &handle) != RTX_OK)
            throw std::runtime_error(std::string("rtx_scene_create: ") + rtx_last_error());
    }
};

// One GpuScene per (Scene, root node), built on first use: the reference's inline intersect functions take raw
// Scene* / Node* on every call, so the mirror keeps the device copies in a registry keyed by those pointers.
// Call ReleaseGpuScenes() before the scene is modified or destroyed.
inline std::map<std::pair<const Scene*, const Node*>, std::unique_ptr<GpuScene>>& GpuSceneRegistry() {
    static std::map<std::pair<const Scene*, const Node*>, std::unique_ptr<GpuScene>> reg;
    return reg;
}
inline GpuScene& GetGpuScene(const Scene* scene, const Node* root = nullptr) {
    auto key = std::make_pair(scene, root ? root : &scene->rootNode);
    auto& reg = GpuSceneRegistry();
    auto it = reg.find(key);
    if (it == reg.end()) it = reg.emplace(key, std::make_unique<GpuScene>(*scene, key.second)).first;
    return *it->second;
}
// Counts the calls of ReleaseGpuScenes(): whoever keeps a handle that a GpuScene owns (PhotonMapBuilder keeps the photon maps
// of a GPU group) compares it with the value it saw when it took the handle.
inline unsigned long& GpuSceneGeneration() {
    static unsigned long gen = 0;
    return gen;
}
inline void ReleaseGpuScenes() {
    ++GpuSceneGeneration();
    GpuSceneRegistry().clear();
}

}  // namespace rtamd
