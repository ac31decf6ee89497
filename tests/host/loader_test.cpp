// Prints what SceneLoader reads from a scene file as JSON (used by tests/test_loader_cpu.py; host code only).
#include <cstdio>
#include "../../raytracer_amd/host/SceneLoader.h"
#include "../../raytracer_amd/host/SceneAdapter.h"
using namespace rtamd;

static void color(const char* k, Color c, const char* end = ",") { std::printf("\"%s\": [%.9g, %.9g, %.9g]%s", k, c.r, c.g, c.b, end); }

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    SceneLoader loader;
    loader.allowMissingMeshes = argc > 2 && std::string(argv[2]) == "--allow-missing";
    LoadedScene ls;
    std::string err;
    if (!loader.Load(argv[1], ls, err)) {
        std::printf("{\"error\": \"%s\"}\n", err.c_str());
        return 1;
    }
    std::vector<SceneNodeItem> items;
    CollectSceneNodeItems(ls.scene, items);
    std::printf("{\"objects\": [");
    bool first = true;
    for (auto const& it : items) {
        const Object* o = it.node->GetNodeObj();
        auto tri = dynamic_cast<const TriObj*>(o);
        const char* kind = tri ? "mesh" : (dynamic_cast<const Sphere*>(o) ? "sphere" : (dynamic_cast<const Plane*>(o) ? "plane" : "none"));
        std::string mname;
        for (auto const& kv : ls.materials)
            if (kv.second.get() == it.node->GetMaterial()) mname = kv.first;
        std::printf("%s{\"kind\": \"%s\", \"nf\": %d, \"nv\": %d, \"nvn\": %d, \"nvt\": %d, \"material\": \"%s\", \"world\": [", first ? "" : ", ",
                    kind, tri ? (int)tri->NF() : 0, tri ? (int)tri->v.size() : 0, tri ? (int)tri->vn.size() : 0, tri ? (int)tri->vt.size() : 0,
                    mname.c_str());
        for (int k = 0; k < 12; ++k) std::printf("%.9g%s", it.world.m[k], k < 11 ? ", " : "");
        std::printf("], \"face_mtl\": [");
        if (tri)
            for (size_t k = 0; k < tri->faceMtl.size(); ++k) std::printf("%d%s", tri->faceMtl[k], k + 1 < tri->faceMtl.size() ? ", " : "");
        std::printf("], \"faces\": [");
        if (tri)
            for (size_t k = 0; k < tri->f.size() && k < 16; ++k)
                std::printf("[%d, %d, %d]%s", tri->f[k].v[0], tri->f[k].v[1], tri->f[k].v[2], (k + 1 < tri->f.size() && k + 1 < 16) ? ", " : "");
        std::printf("]}");
        first = false;
    }
    std::printf("], \"materials\": {");
    first = true;
    for (auto const& kv : ls.materials) {
        const Material& m = *kv.second;
        std::printf("%s\"%s\": {\"type\": %d, ", first ? "" : ", ", kv.first.c_str(), m.Type());
        color("diffuse", m.diffuse.value); color("specular", m.specular.value); color("reflection", m.reflection.value);
        color("refraction", m.refraction.value); color("emission", m.emission.value); color("absorption", m.absorption);
        std::printf("\"glossiness\": %.9g, \"ior\": %.9g, \"bump\": %s, \"bump_value\": %.9g, \"diffuse_tex\": %s, \"diffuse_tex_itm\": [", m.glossiness,
                    m.ior, m.bump ? "true" : "false", m.bumpIntensity, m.diffuse.map ? "true" : "false");
        for (int k = 0; k < 12; ++k) std::printf("%.9g%s", m.diffuse.map ? m.diffuse.map->itm.m[k] : 0.0f, k < 11 ? ", " : "");
        std::printf("], \"diffuse_tex_size\": [%d, %d]}", m.diffuse.map ? m.diffuse.map->width : 0, m.diffuse.map ? m.diffuse.map->height : 0);
        first = false;
    }
    std::printf("}, \"lights\": [");
    for (size_t i = 0; i < ls.scene.lights.size(); ++i) {
        auto L = static_cast<PointLight*>(ls.scene.lights[i]);
        std::printf("%s{", i ? ", " : "");
        color("intensity", L->intensity);
        std::printf("\"position\": [%.9g, %.9g, %.9g], \"size\": %.9g, \"attenuation\": %.9g}", L->position.x, L->position.y, L->position.z, L->size,
                    L->attenuation);
    }
    std::printf("], \"environment\": {");
    color("value", ls.scene.environment.value);
    std::printf("\"tex\": %s, \"itm\": [", ls.scene.environment.map ? "true" : "false");
    for (int k = 0; k < 12; ++k) std::printf("%.9g%s", ls.scene.environment.map ? ls.scene.environment.map->itm.m[k] : 0.0f, k < 11 ? ", " : "");
    const Camera& c = ls.camera;
    std::printf("]}, \"camera\": {\"pos\": [%.9g, %.9g, %.9g], \"dir\": [%.9g, %.9g, %.9g], \"up\": [%.9g, %.9g, %.9g], \"fov\": %.9g, "
                "\"focaldist\": %.9g, \"dof\": %.9g, \"width\": %d, \"height\": %d, \"srgb\": %s, \"keyframes\": [",
                c.pos.x, c.pos.y, c.pos.z, c.dir.x, c.dir.y, c.dir.z, c.up.x, c.up.y, c.up.z, c.fov, c.focaldist, c.dof, c.imgWidth, c.imgHeight,
                c.sRGB ? "true" : "false");
    for (size_t i = 0; i < c.keyframes.size(); ++i)
        std::printf("%s{\"time\": %.9g, \"pos\": [%.9g, %.9g, %.9g], \"target\": [%.9g, %.9g, %.9g], \"fov\": %.9g}", i ? ", " : "",
                    c.keyframes[i].time, c.keyframes[i].pos.x, c.keyframes[i].pos.y, c.keyframes[i].pos.z, c.keyframes[i].target.x,
                    c.keyframes[i].target.y, c.keyframes[i].target.z, c.keyframes[i].fov);
    // Camera::EvaluateAtTime at the times given after the scene file ("--at t0 t1 ..."), as the frame loop would call it
    std::printf("], \"evaluated\": [");
    bool firstT = true;
    for (int a = 2; a < argc; ++a) {
        if (std::string(argv[a]) != "--at") continue;
        Camera cam = c;
        for (int b = a + 1; b < argc; ++b) {
            cam.EvaluateAtTime(std::atof(argv[b]));
            std::printf("%s{\"t\": %s, \"pos\": [%.9g, %.9g, %.9g], \"dir\": [%.9g, %.9g, %.9g], \"up\": [%.9g, %.9g, %.9g], \"fov\": %.9g}",
                        firstT ? "" : ", ", argv[b], cam.pos.x, cam.pos.y, cam.pos.z, cam.dir.x, cam.dir.y, cam.dir.z, cam.up.x, cam.up.y, cam.up.z,
                        cam.fov);
            firstT = false;
        }
        break;
    }
    std::printf("]}, \"warnings\": %d}\n", (int)ls.warnings.size());
    return 0;
}
