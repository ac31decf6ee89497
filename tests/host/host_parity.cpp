// Parity test of the C++ host mirror (raytracer_amd/host/*.h): builds a node tree with the reference's interface,
// runs IntersectNode / IntersectSceneWithLights / TraceRay / TraceShadowRay / RayCaster::CastRays /
// PhotonMapBuilder::BuildPhotonMap on the GPU and checks them against the CPU oracle (oracle/liboracle.so), which
// is linked here as the checker only.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
#include "../../raytracer_amd/host/RayCaster.h"
#include "../../raytracer_amd/host/Animation.h"
#include "../../raytracer_amd/host/ShadeInfoAdapter.h"
#include "../../raytracer_amd/host/Denoiser.h"
#include "../../raytracer_amd/host/SceneLoader.h"

using namespace rtamd;

extern "C" {
int orc_scene_create(const rtx_scene_desc* d, void** out);
void orc_scene_destroy(void* s);
int orc_trace_closest(void* s, const rtx_ray* rays, size_t n, int hit_side, int with_lights, int use_bvh, rtx_hit* hits, uint64_t* counters);
int orc_render(void* s, const rtx_camera* cam, const rtx_render_params* rp, void* photons, int nthreads, int subset_stride, float* accum,
               int* count, float* z, float* albedo, float* normal, float* linear, unsigned char* rgb8, uint64_t* counters, double* secs);
int orc_median_denoise(const float* in, int W, int H, int srgb, float* linear, unsigned char* rgb8);
int orc_trace_secondary(void* s, void* photons, const rtx_shade_point* at, const float* dirs, size_t n, int recursive, int child,
                        uint64_t seed, float* out);
}

static int failures = 0;
#define CHECK(cond, ...)                                   \
    do {                                                   \
        if (!(cond)) {                                     \
            std::printf("FAIL %s:%d: ", __FILE__, __LINE__); \
            std::printf(__VA_ARGS__);                      \
            std::printf("\n");                             \
            ++failures;                                    \
        }                                                  \
    } while (0)

static std::shared_ptr<TriObj> MakeTorus(float R, float r, int nu, int nv) {
    auto t = std::make_shared<TriObj>();
    for (int i = 0; i <= nu; ++i)
        for (int j = 0; j <= nv; ++j) {
            float a = 6.2831853f * i / nu, b = 6.2831853f * j / nv;
            t->v.push_back(Vec3f((R + r * std::cos(b)) * std::cos(a), r * std::sin(b), (R + r * std::cos(b)) * std::sin(a)));
            t->vn.push_back(Vec3f(std::cos(b) * std::cos(a), std::sin(b), std::cos(b) * std::sin(a)));
            t->vt.push_back(Vec3f((float)i / nu, (float)j / nv, 0));
        }
    for (int i = 0; i < nu; ++i)
        for (int j = 0; j < nv; ++j) {
            int a = i * (nv + 1) + j, b = (i + 1) * (nv + 1) + j, c = b + 1, d = a + 1;
            t->f.push_back({{c, b, a}});
            t->f.push_back({{d, c, a}});
        }
    t->fn = t->f;
    t->ft = t->f;
    t->BuildAcceleration();
    return t;
}

int main() {
    if (rtx_device_count() <= 0) {
        std::printf("no HIP device\n");
        return 2;
    }
    // scene graph: root -> {torus A (identity), group (translated) -> {torus B (scaled), glass torus C}}, floor
    Scene scene;
    auto red = std::make_shared<MtlBlinn>();
    red->diffuse.value = Color(0.7f, 0.1f, 0.1f);
    auto glass = std::make_shared<MtlBlinn>();
    glass->diffuse.value = Color(0, 0, 0);
    glass->specular.value = Color(0, 0, 0);
    glass->refraction.value = Color(1, 1, 1);
    glass->glossiness = 1e5f;
    glass->absorption = Color(0.75f, 0.15f, 0.15f);
    auto white = std::make_shared<MtlBlinn>();
    white->diffuse.value = Color(0.8f, 0.8f, 0.8f);
    auto torus = MakeTorus(1.0f, 0.35f, 48, 24);
    Node* a = new Node();
    a->SetNodeObj(torus);
    a->SetMaterial(red);
    scene.rootNode.AppendChild(a);
    Node* grp = new Node();
    grp->tm.m[3] = 2.5f;
    grp->tm.m[7] = 0.3f;
    scene.rootNode.AppendChild(grp);
    Node* b = new Node();
    b->SetNodeObj(torus);
    b->SetMaterial(white);
    b->tm.m[0] = 0.5f; b->tm.m[5] = 0.5f; b->tm.m[10] = 0.5f;
    grp->AppendChild(b);
    Node* c = new Node();
    c->SetNodeObj(torus);
    c->SetMaterial(glass);
    c->tm.m[7] = 1.2f;
    grp->AppendChild(c);
    auto floorObj = std::make_shared<TriObj>();
    floorObj->v = {Vec3f(-8, -0.4f, -8), Vec3f(8, -0.4f, -8), Vec3f(8, -0.4f, 8), Vec3f(-8, -0.4f, 8)};
    floorObj->f = {{{0, 2, 1}}, {{0, 3, 2}}};
    floorObj->BuildAcceleration();
    Node* fl = new Node();
    fl->SetNodeObj(floorObj);
    fl->SetMaterial(white);
    scene.rootNode.AppendChild(fl);
    // analytic objects (objects.cpp:21-126): an ellipsoid and a tilted square mirror
    Node* sp = new Node();
    sp->SetNodeObj(std::make_shared<Sphere>());
    sp->SetMaterial(red);
    sp->tm.m[0] = 0.6f; sp->tm.m[5] = 0.4f; sp->tm.m[10] = 0.5f;
    sp->tm.m[3] = -1.8f; sp->tm.m[7] = 0.6f; sp->tm.m[11] = 1.4f;
    scene.rootNode.AppendChild(sp);
    Node* pl = new Node();
    pl->SetNodeObj(std::make_shared<Plane>());
    pl->SetMaterial(white);
    pl->tm.m[0] = 1.5f; pl->tm.m[5] = 0.0f; pl->tm.m[6] = -1.5f; pl->tm.m[9] = 1.5f; pl->tm.m[10] = 0.0f;   // rotate 90 deg about x, scale 1.5
    pl->tm.m[3] = 0.5f; pl->tm.m[7] = 1.0f; pl->tm.m[11] = -3.0f;
    scene.rootNode.AppendChild(pl);
    PointLight* L = scene.AddPointLight();
    L->position = Vec3f(1, 6, 2);
    L->intensity = Color(20, 18, 15);
    L->size = 0.8f;
    scene.environment.value = Color(0.3f, 0.35f, 0.4f);

    // CollectSceneNodeObjects: DFS pre-order (SceneAdapter.h:8-18)
    std::vector<std::pair<const Node*, const Object*>> no;
    CollectSceneNodeObjects(scene, no);
    CHECK(no.size() == 6 && no[0].first == a && no[1].first == b && no[2].first == c && no[3].first == fl && no[4].first == sp &&
              no[5].first == pl,
          "DFS order");

    // the same flattened description for the oracle
    std::vector<rtx_mesh_desc> meshes;
    std::vector<rtx_material_desc> mtls(3);
    auto setm = [](rtx_material_desc& d, const Material& m) {
        d = rtx_material_desc{};
        d.type = m.Type();
        const Color* src[5] = {&m.diffuse.value, &m.specular.value, &m.reflection.value, &m.refraction.value, &m.emission.value};
        float* dst[5] = {d.diffuse, d.specular, d.reflection, d.refraction, d.emission};
        for (int k = 0; k < 5; ++k) { dst[k][0] = src[k]->r; dst[k][1] = src[k]->g; dst[k][2] = src[k]->b; }
        d.diffuse_tex = d.specular_tex = d.reflection_tex = d.refraction_tex = d.emission_tex = d.bump_tex = d.glossiness_tex = -1;
        d.glossiness = m.glossiness; d.ior = m.ior; d.bump_scale = 1;
        d.absorption[0] = m.absorption.r; d.absorption[1] = m.absorption.g; d.absorption[2] = m.absorption.b;
    };
    setm(mtls[0], *red); setm(mtls[1], *white); setm(mtls[2], *glass);
    const Matrix34 worlds[4] = {a->tm, Matrix34::Compose(grp->tm, b->tm), Matrix34::Compose(grp->tm, c->tm), fl->tm};
    const int mi[4] = {0, 1, 2, 1};
    for (int k = 0; k < 4; ++k) {
        auto t = static_cast<const TriObj*>(no[k].second);
        rtx_mesh_desc d{};
        d.v = &t->v.data()->x; d.nv = (int)t->v.size();
        d.vn = t->vn.empty() ? nullptr : &t->vn.data()->x; d.nvn = (int)t->vn.size();
        d.vt = t->vt.empty() ? nullptr : &t->vt.data()->x; d.nvt = (int)t->vt.size();
        d.f = t->f.data()->v; d.nf = (int)t->f.size();
        for (int q = 0; q < 12; ++q) d.xform[q] = worlds[k].m[q];
        d.material = mi[k];
        meshes.push_back(d);
    }
    for (int k = 4; k < 6; ++k) {
        rtx_mesh_desc d{};
        d.kind = k == 4 ? RTX_OBJ_SPHERE : RTX_OBJ_PLANE;
        const Matrix34& w = k == 4 ? sp->tm : pl->tm;
        for (int q = 0; q < 12; ++q) d.xform[q] = w.m[q];
        d.material = k == 4 ? 0 : 1;
        meshes.push_back(d);
    }
    rtx_light_desc ld{{1, 6, 2}, {20, 18, 15}, 0.8f, 1.0f};
    rtx_scene_desc sd{};
    sd.meshes = meshes.data(); sd.n_meshes = 6;
    sd.materials = mtls.data(); sd.n_materials = 3;
    sd.lights = &ld; sd.n_lights = 1;
    sd.environment[0] = 0.3f; sd.environment[1] = 0.35f; sd.environment[2] = 0.4f;
    sd.environment_tex = sd.background_tex = -1;
    void* osc = nullptr;
    orc_scene_create(&sd, &osc);

    // IntersectNode / IntersectSceneWithLights per ray and batched, all hit sides
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> U(-1, 1);
    const int N = 4000;
    std::vector<Ray> rays(N);
    std::vector<rtx_ray> rr(N);
    for (int i = 0; i < N; ++i) {
        Vec3f o(4 * U(rng) + 1, 3 * U(rng) + 1, 4 * U(rng)), t(2 * U(rng) + 1, U(rng), 2 * U(rng));
        rays[i] = Ray(o, (t - o).GetNormalized());
        rr[i] = detail::ToRay(rays[i], BIGFLOAT);
    }
    for (int side = HIT_FRONT; side <= HIT_FRONT_AND_BACK; ++side) {
        std::vector<rtx_hit> oh(N);
        orc_trace_closest(osc, rr.data(), N, side, 1, 1, oh.data(), nullptr);
        std::vector<ExtendedHitInfo> hits(N);
        for (auto& h : hits) h.Init();
        std::vector<bool> flags;
        IntersectSceneBatch(&scene, rays, hits, side, true, &flags);
        int mism = 0, nhit = 0;
        for (int i = 0; i < N; ++i) {
            bool ohit = oh[i].obj >= 0 || oh[i].light >= 0;
            if (flags[i] != ohit) { ++mism; continue; }
            if (!ohit) continue;
            ++nhit;
            const Node* expect = oh[i].obj >= 0 ? no[oh[i].obj].first : nullptr;
            if (hits[i].node != expect || (hits[i].hitLight != nullptr) != (oh[i].light >= 0)) ++mism;
            else if (std::fabs(hits[i].z - oh[i].t) > 1e-4f * oh[i].t) ++mism;
        }
        CHECK(mism <= N / 500 && nhit > N / 10, "side %d: %d mismatches of %d (%d hits)", side, mism, N, nhit);
    }
    {   // single-ray entry points
        ExtendedHitInfo e; e.Init();
        rtx_hit oh;
        orc_trace_closest(osc, &rr[0], 1, HIT_FRONT_AND_BACK, 1, 1, &oh, nullptr);
        bool hit = IntersectSceneWithLights(&scene, rays[0], e, HIT_FRONT_AND_BACK);
        CHECK(hit == (oh.obj >= 0 || oh.light >= 0), "IntersectSceneWithLights hit flag");
        HitInfo h; h.Init();
        bool hn = IntersectNode(&scene.rootNode, rays[1], h, HIT_FRONT_AND_BACK);
        orc_trace_closest(osc, &rr[1], 1, HIT_FRONT_AND_BACK, 0, 1, &oh, nullptr);
        CHECK(hn == (oh.obj >= 0) && (!hn || std::fabs(h.z - oh.t) <= 1e-4f * oh.t), "IntersectNode");
        // in-out bound (IntersectNodeAdapter.h:103): a bound below the hit distance means no hit
        if (hn) {
            HitInfo h2; h2.Init(); h2.z = h.z * 0.5f;
            CHECK(!IntersectNode(&scene.rootNode, rays[1], h2, HIT_FRONT_AND_BACK) && h2.z == h.z * 0.5f, "bound in-out");
        }
        // sub-tree query: IntersectNode on the group node sees only B and C
        HitInfo hg; hg.Init();
        Ray down(Vec3f(2.5f + 0.5f, 5, 0), Vec3f(0, -1, 0));
        CHECK(IntersectNode(grp, down, hg, HIT_FRONT_AND_BACK) && (hg.node == b || hg.node == c), "sub-tree IntersectNode");
        HitInfo ht;
        CHECK(TraceRay(scene, down, ht) && ht.node == hg.node && std::fabs(ht.z - hg.z) < 1e-4f, "TraceRay");
        CHECK(TraceShadowRay(scene, down, ht.z * 0.5f) && !TraceShadowRay(scene, down, ht.z * 2.0f), "TraceShadowRay");
        ShadeInfoAdapter si(scene.lights, &scene);
        HitInfo fh; fh.Init(); fh.p = Vec3f(2.5f + 0.5f, -0.4f, 0); fh.N = Vec3f(0, 1, 0); fh.node = fl;
        si.SetHit(Ray(Vec3f(0, 3, 0), Vec3f(0, -1, 0)), fh);
        CHECK(si.TraceShadowRay(Vec3f(0, 1, 0), 10.0f) == 0.0f, "legacy shadow ray is blocked by the tori above");
        CHECK(si.TraceShadowRay(Vec3f(0, 1, 0), 0.2f) == 1.0f, "legacy shadow ray: segment ends before the blocker");
    }

    // RayCaster::CastRays against the oracle render (fixed 8 spp), then adaptive, then Stop()
    Camera cam;
    cam.pos = Vec3f(0.5f, 2.5f, 6.5f);
    cam.dir = (Vec3f(1.2f, 0.3f, 0) - cam.pos).GetNormalized();
    Vec3f x = cam.dir ^ Vec3f(0, 1, 0);
    cam.up = (x ^ cam.dir).GetNormalized();
    cam.fov = 35; cam.focaldist = 6.5f; cam.dof = 0.03f; cam.imgWidth = 96; cam.imgHeight = 64; cam.sRGB = true;
    RayCaster rc(&scene, &cam);
    rc.sampling.minSamples = rc.sampling.maxSamples = 8;
    RenderImage img;
    img.Init(cam.imgWidth, cam.imgHeight);
    rc.CastRays(img);
    rtx_camera ocam{};
    ocam.pos[0] = cam.pos.x; ocam.pos[1] = cam.pos.y; ocam.pos[2] = cam.pos.z;
    ocam.dir[0] = cam.dir.x; ocam.dir[1] = cam.dir.y; ocam.dir[2] = cam.dir.z;
    ocam.up[0] = cam.up.x; ocam.up[1] = cam.up.y; ocam.up[2] = cam.up.z;
    ocam.fov = cam.fov; ocam.focaldist = cam.focaldist; ocam.dof = cam.dof; ocam.width = 96; ocam.height = 64; ocam.srgb = 1;
    rtx_render_params rp{};
    rp.spp_min = rp.spp_max = 8; rp.index_stride = 1024; rp.ci_halfwidth = 0.005f; rp.filter_type = 0; rp.filter_radius = 1.0f;
    rp.max_bounces = 64; rp.seed = 0x9e3779b97f4a7c15ULL; rp.tile_size = 16; rp.world = 1;
    const size_t np = 96 * 64;
    std::vector<float> lin(np * 3), z(np);
    std::vector<int> cnt(np);
    std::vector<unsigned char> rgb(np * 3);
    orc_render(osc, &ocam, &rp, nullptr, 4, 1, nullptr, cnt.data(), z.data(), nullptr, nullptr, lin.data(), rgb.data(), nullptr, nullptr);
    double se = 0, mean = 0;
    int zbad = 0;
    for (size_t i = 0; i < np; ++i) {
        const Color& g = img.GetLinearPixels()[i];
        se += std::pow(g.r - lin[3 * i], 2) + std::pow(g.g - lin[3 * i + 1], 2) + std::pow(g.b - lin[3 * i + 2], 2);
        mean += lin[3 * i] + lin[3 * i + 1] + lin[3 * i + 2];
        if (std::fabs(img.GetZBuffer()[i] - z[i]) > 1e-4f * std::fabs(z[i])) ++zbad;
        if (img.GetSampleCount()[i] != cnt[i]) ++zbad;
    }
    double rmse = std::sqrt(se / (3 * np));
    mean /= 3 * np;
    CHECK(rmse <= 0.02 * mean && zbad <= (int)np / 1000, "CastRays: rmse %.5f mean %.5f zbad %d", rmse, mean, zbad);
    CHECK(img.GetNumRenderedPixels() == (int)np, "rendered pixel counter");

    // legacy shading mode (SHADING_USE_LEGACY=1): the same frame through TraceLegacy, and ShadeInfoAdapter's secondary ray
    {
        RayCaster rl(&scene, &cam);
        rl.sampling.minSamples = rl.sampling.maxSamples = 4;
        rl.sampling.legacyShading = true;
        RenderImage imgL;
        imgL.Init(cam.imgWidth, cam.imgHeight);
        rl.CastRays(imgL);
        rtx_render_params rl_p = rp;
        rl_p.spp_min = rl_p.spp_max = 4;
        rl_p.shading_mode = RTX_SHADING_LEGACY;
        std::vector<float> linL(np * 3);
        orc_render(osc, &ocam, &rl_p, nullptr, 4, 1, nullptr, nullptr, nullptr, nullptr, nullptr, linL.data(), nullptr, nullptr, nullptr);
        double seL = 0, meanL = 0, diffModes = 0;
        for (size_t i = 0; i < np; ++i) {
            const Color& g = imgL.GetLinearPixels()[i];
            seL += std::pow(g.r - linL[3 * i], 2) + std::pow(g.g - linL[3 * i + 1], 2) + std::pow(g.b - linL[3 * i + 2], 2);
            meanL += linL[3 * i] + linL[3 * i + 1] + linL[3 * i + 2];
            diffModes += std::fabs(g.r - img.GetLinearPixels()[i].r);
        }
        const double rmseL = std::sqrt(seL / (3 * np));
        meanL /= 3 * np;
        CHECK(rmseL <= 0.02 * meanL, "legacy CastRays: rmse %.5f mean %.5f", rmseL, meanL);
        if (rmseL > 0.02 * meanL) {   // where the frames differ most
            std::vector<std::pair<float, size_t>> worst;
            for (size_t i = 0; i < np; ++i) {
                const Color& g = imgL.GetLinearPixels()[i];
                worst.push_back({std::max(std::fabs(g.r - linL[3 * i]), std::max(std::fabs(g.g - linL[3 * i + 1]), std::fabs(g.b - linL[3 * i + 2]))), i});
            }
            std::sort(worst.rbegin(), worst.rend());
            int nbig = 0;
            for (auto& w : worst) nbig += w.first > 0.01f;
            std::printf("  pixels off by > 0.01: %d\n", nbig);
            for (int k = 0; k < 8; ++k) {
                const size_t i = worst[k].second;
                const Color& g = imgL.GetLinearPixels()[i];
                std::printf("  (%3zu,%3zu) gpu %.4f %.4f %.4f  oracle %.4f %.4f %.4f\n", i % 96, i / 96, g.r, g.g, g.b, linL[3 * i], linL[3 * i + 1], linL[3 * i + 2]);
            }
        }
        CHECK(diffModes / np > 1e-3, "legacy and path frames differ");

        ShadeInfoAdapter si(scene.lights, &scene);
        HitInfo fh; fh.Init(); fh.p = Vec3f(0.2f, 0.0f, 0.3f); fh.N = Vec3f(0, 1, 0); fh.node = nullptr;
        si.SetHit(Ray(Vec3f(0, 3, 0), Vec3f(0, -1, 0)), fh);
        si.SetPixel(5, 7); si.SetPixelSample(2); si.imageWidth = cam.imgWidth; si.bounce = 1; si.rngNode = 1;
        rtx_shade_point at{};
        at.P[0] = fh.p.x; at.P[1] = fh.p.y; at.P[2] = fh.p.z; at.N[1] = 1.0f;
        at.obj = -1; at.bounce = 1; at.diffuse_depth = 0; at.pixel = 7 * cam.imgWidth + 5; at.sample = 2; at.node = 1;
        int bad = 0, hits = 0;
        std::mt19937 gen(3);
        std::uniform_real_distribution<float> U(-1.0f, 1.0f);
        for (int k = 0; k < 64; ++k) {
            const Vec3f d(U(gen), std::fabs(U(gen)) + 0.05f, U(gen));
            float dist = 0, ref[4];
            const Color col = si.TraceSecondaryRay(d, dist, true);
            const float dd[3] = {d.x, d.y, d.z};
            orc_trace_secondary(osc, nullptr, &at, dd, 1, 1, 2, si.seed, ref);
            const float tol = 2e-3f * (1.0f + std::max(ref[0], std::max(ref[1], ref[2])));
            if (std::fabs(col.r - ref[0]) > tol || std::fabs(col.g - ref[1]) > tol || std::fabs(col.b - ref[2]) > tol) ++bad;
            if ((dist > 1e29f) != (ref[3] > 1e29f) || (ref[3] < 1e29f && std::fabs(dist - ref[3]) > 1e-4f * (1 + ref[3]))) ++bad;
            if (ref[3] < 1e29f) ++hits;
        }
        CHECK(bad <= 1 && hits > 5, "TraceSecondaryRay vs oracle: %d of 64 differ (%d hit geometry)", bad, hits);
        si.bounce = 5;
        float dist = 0;
        const Color none = si.TraceSecondaryRay(Vec3f(0, 1, 0), dist);
        CHECK(none.r == 0 && none.g == 0 && none.b == 0 && dist > 1e29f, "CanBounce stops the recursion");
    }

    // photon map through the mirrored builder
    PhotonMapBuilder pmb;
    pmb.SetNumPhotons(200000);
    pmb.BuildPhotonMap(&scene);
    CHECK(pmb.GetPhotonMap()->NumPhotons() > 0, "global photon map is populated");
    CHECK(pmb.GetCausticSearchRadius() >= 0.01f && pmb.GetPhotonSearchRadius() >= 0.05f, "gather radii");
    RenderImage img2;
    img2.Init(cam.imgWidth, cam.imgHeight);
    rc.CastRays(img2, &pmb);

    // outputs: median denoise on the GPU against the oracle, PNG files of colour / z / sample counts
    {
        std::vector<unsigned char> gpu8(np * 3), cpu8(np * 3);
        std::string err;
        CHECK(Denoiser::ApplyMedian(img, true, "/tmp/rtamd_denoised.png", reinterpret_cast<Color24*>(gpu8.data()), &err), "ApplyMedian: %s", err.c_str());
        orc_median_denoise(&img.GetLinearPixels()->r, cam.imgWidth, cam.imgHeight, 1, nullptr, cpu8.data());
        int off = 0;
        for (size_t i = 0; i < np * 3; ++i) off += std::abs((int)gpu8[i] - (int)cpu8[i]) > 1;
        CHECK(off == 0, "median denoise: %d bytes differ by more than one level", off);
        img.ComputeZBufferImage();
        const int smax = img.ComputeSampleCountImage();
        CHECK(smax == 8, "ComputeSampleCountImage returns the maximum count (%d)", smax);
        CHECK(img.SaveImage("/tmp/rtamd_render.png") && img.SaveZImage("/tmp/rtamd_z.png") && img.SaveSampleCountImage("/tmp/rtamd_samples.png"), "PNG outputs");
        FILE* f = std::fopen("/tmp/rtamd_render.png", "rb");
        unsigned char sig[8] = {0};
        CHECK(f && std::fread(sig, 1, 8, f) == 8 && sig[1] == 'P' && sig[2] == 'N' && sig[3] == 'G', "PNG signature");
        if (f) std::fclose(f);
    }

    // progressive preview (RayCaster.cpp:1215-1279): 1-spp visits of bucket batches until every pixel holds maxSamples;
    // the same samples as the fixed 8-spp frame above, so the same image up to the order of the film's float atomics
    {
        RayCaster rcp(&scene, &cam);
        rcp.sampling.minSamples = rcp.sampling.maxSamples = 8;
        rcp.sampling.maxBounces = rc.sampling.maxBounces;
        rcp.SetProgressivePreview(true);
        RenderImage imgP;
        imgP.Init(cam.imgWidth, cam.imgHeight);
        rcp.CastRays(imgP);
        int badCount = 0, badPix = 0;
        for (size_t i = 0; i < np; ++i) {
            if (imgP.GetSampleCount()[i] != 8) ++badCount;
            const Color a = imgP.GetLinearPixels()[i], b = img.GetLinearPixels()[i];
            if (std::fabs(a.r - b.r) > 2e-4f * (1 + b.r) || std::fabs(a.g - b.g) > 2e-4f * (1 + b.g) || std::fabs(a.b - b.b) > 2e-4f * (1 + b.b)) ++badPix;
        }
        CHECK(badCount == 0 && badPix == 0, "progressive preview: %d sample counts off, %d pixels differ", badCount, badPix);
        CHECK(imgP.GetNumRenderedPixels() >= (int)np, "progressive preview advances the rendered-pixel counter");
    }

    // scene ingest -> GPU -> image: an XML scene with an OBJ mesh, a sphere and a plane, rendered through RayCaster
    {
        FILE* f = std::fopen("/tmp/rtamd_scene.xml", "w");
        std::fputs("<xml><scene>"
                   "<object type='obj' name='rtamd_tri.obj' material='red'><translate x='0' y='0.5' z='0'/></object>"
                   "<object type='sphere' name='ball' material='mirror'><scale value='0.6'/><translate x='1.4' y='0.6' z='0'/></object>"
                   "<object type='plane' name='floor' material='white'><rotate angle='-90' x='1'/><scale value='5'/></object>"
                   "<material type='blinn' name='red'><diffuse r='0.8' g='0.1' b='0.1'/><specular value='0.2'/></material>"
                   "<material type='blinn' name='mirror'><diffuse value='0.05'/><reflection value='0.9'/><glossiness value='2000'/></material>"
                   "<material type='blinn' name='white'><diffuse value='0.7'/><specular value='0'/></material>"
                   "<light type='point' name='l'><intensity value='30'/><position x='2' y='5' z='3'/><size value='0.5'/></light>"
                   "<environment r='0.2' g='0.25' b='0.3'/>"
                   "</scene><camera><position x='0.5' y='1.5' z='5'/><target x='0.5' y='0.5' z='0'/><up x='0' y='1' z='0'/>"
                   "<fov value='40'/><width value='80'/><height value='48'/></camera></xml>", f);
        std::fclose(f);
        f = std::fopen("/tmp/rtamd_tri.obj", "w");
        std::fputs("v -1 0 0\nv 0 0 0\nv -0.5 1 0\nvn 0 0 1\nf 1//1 2//1 3//1\n", f);
        std::fclose(f);
        SceneLoader loader;
        LoadedScene ls;
        std::string err;
        CHECK(loader.Load("/tmp/rtamd_scene.xml", ls, err), "SceneLoader: %s", err.c_str());
        RayCaster rcx(&ls.scene, &ls.camera);
        rcx.sampling.minSamples = rcx.sampling.maxSamples = 4;
        rcx.sampling.maxBounces = 4;
        RenderImage imx;
        imx.Init(ls.camera.imgWidth, ls.camera.imgHeight);
        rcx.CastRays(imx);
        double meanx = 0;
        for (int i = 0; i < 80 * 48; ++i) meanx += imx.GetLinearPixels()[i].r + imx.GetLinearPixels()[i].g + imx.GetLinearPixels()[i].b;
        CHECK(meanx / (80 * 48 * 3) > 0.02, "loaded scene renders (mean %.4f)", meanx / (80 * 48 * 3));
        HitInfo hs;
        hs.Init();
        Ray toBall(ls.camera.pos, (Vec3f(1.4f, 0.6f, 0.0f) - ls.camera.pos).GetNormalized());
        CHECK(IntersectNode(&ls.scene.rootNode, toBall, hs, HIT_FRONT) && dynamic_cast<const Sphere*>(hs.node->GetNodeObj()), "ray to the sphere hits it");
        const float expectT = (Vec3f(1.4f, 0.6f, 0.0f) - ls.camera.pos).Length() - 0.6f;
        CHECK(std::fabs(hs.z - expectT) < 1e-4f, "sphere hit distance %.6f vs %.6f", hs.z, expectT);
        ReleaseGpuScenes();
    }

    // Stop() from another thread ends a long render early
    rc.sampling.minSamples = rc.sampling.maxSamples = 1024;
    rc.sampling.samplesPerPass = 2;   // 512 passes, so the render outlasts the stopper's delay on any GPU
    RenderImage img3;
    img3.Init(cam.imgWidth, cam.imgHeight);
    std::thread stopper([&] {
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
        rc.Stop();
    });
    rc.CastRays(img3);
    stopper.join();
    CHECK(rc.IsStopped(), "IsStopped after Stop()");
    long total = 0;
    for (size_t i = 0; i < np; ++i) total += img3.GetSampleCount()[i];
    CHECK(total < (long)np * 1024, "cancelled render took fewer samples (%ld)", total);

    // animation frame loop (Main.cpp:381-401): the camera moves along its keyframes, the scene stays on the GPU
    {
        Camera anim = cam;
        anim.imgWidth = 64; anim.imgHeight = 40;
        anim.upHint = Vec3f(0, 1, 0);
        anim.keyframes = {{0.0, cam.pos, Vec3f(1.2f, 0.3f, 0), 35.0f}, {1.0, cam.pos + Vec3f(2.0f, 0.5f, -1.0f), Vec3f(1.2f, 0.3f, 0), 30.0f}};
        RayCaster ra(&scene, &anim);
        ra.sampling.minSamples = ra.sampling.maxSamples = 2;
        RenderImage imgA;
        SequenceOptions so;
        so.frames = 3; so.fps = 2.0; so.start = 0.0; so.end = 1.0;
        so.outDir = "/tmp/rtx_host_parity_seq";
        std::vector<std::vector<Color>> frames;
        std::vector<double> times;
        const size_t scenesBefore = GpuSceneRegistry().size();
        const int nDone = RenderSequence(ra, anim, imgA, nullptr, so, [&](int, double t, RenderImage& im, double) {
            frames.emplace_back(im.GetLinearPixels(), im.GetLinearPixels() + (size_t)im.GetWidth() * im.GetHeight());
            times.push_back(t);
        });
        CHECK(nDone == 3 && times.size() == 3 && times[0] == 0.0 && times[1] == 0.5 && times[2] == 1.0, "frame times of the sequence");
        CHECK(GpuSceneRegistry().size() == scenesBefore, "the scene is not rebuilt between frames");
        double d01 = 0, d12 = 0;
        for (size_t i = 0; i < frames[0].size(); ++i) {
            d01 += std::fabs(frames[0][i].r - frames[1][i].r);
            d12 += std::fabs(frames[1][i].r - frames[2][i].r);
        }
        CHECK(d01 / frames[0].size() > 1e-3 && d12 / frames[0].size() > 1e-3, "the camera moves between frames");
        const OutputNames on = FormatBaseNames(so.outDir, so.prefix, 2, 4);
        CHECK(on.color == "/tmp/rtx_host_parity_seq/beauty/render_0002.png", "output naming: %s", on.color.c_str());
        FILE* fp = std::fopen(on.color.c_str(), "rb");
        CHECK(fp != nullptr, "frame 2 was written");
        if (fp) std::fclose(fp);
    }

    // several GPUs from this one process (SamplingConfig::devices -> rtx_group): two members sharing the test GPU must
    // reproduce the single-device frame - same sample counts and z, colour up to float summation order - with and
    // without the photon map (every member traces half of the photon ids, all hold the whole map afterwards)
    {
        ReleaseGpuScenes();
        RayCaster rg(&scene, &cam);
        rg.sampling.minSamples = rg.sampling.maxSamples = 8;
        rg.sampling.devices = {0, 0};
        RenderImage imgG;
        imgG.Init(cam.imgWidth, cam.imgHeight);
        rg.CastRays(imgG);
        CHECK(GetGpuScene(&scene).Group() != nullptr && rtx_group_size(GetGpuScene(&scene).Group()) == 2, "the scene lives in a 2-member group");
        int badG = 0;
        double dG = 0;
        for (size_t i = 0; i < np; ++i) {
            const Color &a = imgG.GetLinearPixels()[i], &b = img.GetLinearPixels()[i];
            dG = std::max(dG, (double)std::max(std::fabs(a.r - b.r), std::max(std::fabs(a.g - b.g), std::fabs(a.b - b.b))) / (1.0 + b.r + b.g + b.b));
            if (imgG.GetSampleCount()[i] != img.GetSampleCount()[i] || imgG.GetZBuffer()[i] != img.GetZBuffer()[i]) ++badG;
        }
        CHECK(badG == 0 && dG < 2e-4, "group CastRays vs single device: %d counts / z differ, max relative colour difference %.2e", badG, dG);
        PhotonMapBuilder pmg;
        pmg.SetNumPhotons(200000);
        pmg.BuildPhotonMap(&scene);
        CHECK(pmg.GetPhotonMap()->NumPhotons() == pmb.GetPhotonMap()->NumPhotons() && pmg.GetCausticsMap()->NumPhotons() == pmb.GetCausticsMap()->NumPhotons(),
              "group photon map holds the single-device counts (%d / %d vs %d / %d)", pmg.GetPhotonMap()->NumPhotons(),
              pmg.GetCausticsMap()->NumPhotons(), pmb.GetPhotonMap()->NumPhotons(), pmb.GetCausticsMap()->NumPhotons());
        RenderImage imgGP;
        imgGP.Init(cam.imgWidth, cam.imgHeight);
        rg.CastRays(imgGP, &pmg);
        double dP = 0;
        for (size_t i = 0; i < np; ++i) {
            const Color &a = imgGP.GetLinearPixels()[i], &b = img2.GetLinearPixels()[i];
            dP = std::max(dP, (double)std::max(std::fabs(a.r - b.r), std::max(std::fabs(a.g - b.g), std::fabs(a.b - b.b))) / (1.0 + b.r + b.g + b.b));
        }
        CHECK(dP < 1e-3, "group CastRays with photons vs single device: max relative difference %.2e", dP);
        // adaptive frame (8 ... 32 samples): the members deal the unconverged buckets out again after every round
        // (rtx_group.cpp); the sample map must be the single device's
        rg.sampling.minSamples = 8;
        rg.sampling.maxSamples = 32;
        rg.sampling.maxHalfWidth = 0.02f;
        RenderImage imgGA;
        imgGA.Init(cam.imgWidth, cam.imgHeight);
        rg.CastRays(imgGA);
        g_rtDevices.clear();
        ReleaseGpuScenes();
        RayCaster r1(&scene, &cam);
        r1.sampling = rg.sampling;
        r1.sampling.devices.clear();
        RenderImage img1A;
        img1A.Init(cam.imgWidth, cam.imgHeight);
        r1.CastRays(img1A);
        int badA = 0, refined = 0;
        double dA = 0;
        for (size_t i = 0; i < np; ++i) {
            const Color &a = imgGA.GetLinearPixels()[i], &b = img1A.GetLinearPixels()[i];
            dA = std::max(dA, (double)std::max(std::fabs(a.r - b.r), std::max(std::fabs(a.g - b.g), std::fabs(a.b - b.b))) / (1.0 + b.r + b.g + b.b));
            if (imgGA.GetSampleCount()[i] != img1A.GetSampleCount()[i]) ++badA;
            if (img1A.GetSampleCount()[i] > 8) ++refined;
        }
        CHECK(badA == 0 && refined > 0 && dA < 2e-4, "adaptive group CastRays vs single device: %d sample counts differ, %d pixels refined, max relative colour difference %.2e",
              badA, refined, dA);
        ReleaseGpuScenes();
    }

    // RayCastRenderer (RayCaster.h:66-117): LoadScene -> BeginRender on its own thread -> outputs; single-ray queries
    {
        RayCastRenderer rr;
        CHECK(rr.LoadScene("/tmp/rtamd_scene.xml"), "RayCastRenderer::LoadScene");
        rr.GetRayCaster()->sampling.minSamples = rr.GetRayCaster()->sampling.maxSamples = 4;
        rr.GetRayCaster()->sampling.maxBounces = 4;
        rr.SetOutputPaths("/tmp/rtamd_rr_render.png", "/tmp/rtamd_rr_z.png", "/tmp/rtamd_rr_samples.png", "/tmp/rtamd_rr_denoised.png");
        PhotonMapBuilder none;
        CHECK(rr.GetPhotonMap() == nullptr && rr.GetCausticsMap() == nullptr, "no photon mapper set");
        rr.SetPhotonMapper(&none);
        CHECK(rr.GetPhotonMap() == none.GetPhotonMap() && rr.GetCausticsMap() == none.GetCausticsMap(), "photon map getters forward to the mapper");
        rr.SetPhotonMapper(nullptr);
        rr.BeginRender();
        rr.WaitRender();
        CHECK(!rr.IsRendering(), "offline render clears isRendering when done");
        double meanr = 0;
        for (int i = 0; i < 80 * 48; ++i) meanr += rr.GetRenderImage().GetLinearPixels()[i].r;
        CHECK(meanr / (80 * 48) > 0.02, "RayCastRenderer frame (mean %.4f)", meanr / (80 * 48));
        FILE* fr = std::fopen("/tmp/rtamd_rr_render.png", "rb");
        CHECK(fr != nullptr, "RayCastRenderer saved its outputs");
        if (fr) std::fclose(fr);
        HitInfo hr;
        Ray toBall(rr.GetCamera().pos, (Vec3f(1.4f, 0.6f, 0.0f) - rr.GetCamera().pos).GetNormalized());
        const float expectT = (Vec3f(1.4f, 0.6f, 0.0f) - rr.GetCamera().pos).Length() - 0.6f;
        CHECK(rr.TraceRay(toBall, hr) && std::fabs(hr.z - expectT) < 1e-4f, "RayCastRenderer::TraceRay");
        CHECK(!rr.TraceShadowRay(toBall, expectT * 2.0f) && rr.TraceShadowRay(toBall, expectT * 0.5f), "RayCastRenderer::TraceShadowRay");
        RNG rng(7);
        ShadeInfoAdapter sir(rr.GetScene().lights, &rr.GetScene(), rng);   // the reference's constructor (ShadeInfoAdapter.h:21)
        const ShadeInfo& base = sir;
        HitInfo fh; fh.Init(); fh.p = Vec3f(0.5f, 0.0f, 1.0f); fh.N = Vec3f(0, 1, 0); fh.node = nullptr;
        sir.SetHit(Ray(Vec3f(0.5f, 3, 1), Vec3f(0, -1, 0)), fh);
        CHECK(base.TraceShadowRay(Ray(fh.p, Vec3f(0, 1, 0)), BIGFLOAT) == 1.0f, "ShadeInfo::TraceShadowRay through the base class");
        ReleaseGpuScenes();
    }

    ReleaseGpuScenes();
    orc_scene_destroy(osc);
    std::printf(failures ? "host_parity: %d FAILURES\n" : "host_parity: all checks passed\n", failures);
    return failures ? 1 : 0;
}
