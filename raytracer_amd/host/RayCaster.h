// Mirror of the reference's RayTracer/RayCaster.h: RayCaster(Scene*, Camera*) / CastRays / Stop / IsStopped /
// SetProgressivePreview and the RayCastRenderer ray queries, with the same signatures. CastRays blocks until the
// frame is done, fills the RenderImage arrays (RayCaster.cpp:1183-1211) and may be cancelled from another thread.
#pragma once
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <memory>
#include <string>
#include <thread>
#include "IntersectNodeAdapter.h"
#include "PhotonMapBuilder.h"
#include "SceneLoader.h"

namespace rtamd {

struct Bucket {
    int x0, y0, x1, y1;
};
enum class FilterType { Tent, Gaussian, Mitchell, Lanczos };

// RayCaster.cpp:30-31 (globals set from main)
inline FilterType g_filterType = FilterType::Tent;
inline float g_filterRadius = 1.0f;

// The reference's compile-time sampling constants (RayCaster.cpp:51-56, :358) become run-time fields so that the
// fixed-spp configurations of BASELINE.json can be rendered; the defaults are the reference's values.
struct SamplingConfig {
    int minSamples = 128;
    int maxSamples = 1024;
    float maxHalfWidth = 0.005f;
    int maxBounces = 64;
    uint64_t seed = 0x9e3779b97f4a7c15ULL;
    int rank = 0, world = 1;          // tile sharding across GPUs (one process per GPU)
    int samplesPerPass = 0;           // samples per pixel in flight at once; 0 = sized from free HBM
    int progressiveSamplesPerPass = 1;   // kProgressiveSamplesPerPass, RayCaster.cpp:62
    bool legacyShading = false;       // SHADING_USE_LEGACY (ShadingConfig.h:7-13): TraceLegacy instead of TracePath
    std::vector<int> devices;         // HIP devices to render on from this one process; sets g_rtDevices (SceneAdapter.h)
                                      // before the scene is handed to the GPU. Empty: leave g_rtDevices as it is
};

class RayCaster {
public:
    RayCaster(Scene* scene, Camera* camera) : scene(scene), camera(camera) {}

    // RayCaster.cpp:1146-1314
    void CastRays(RenderImage& image, PhotonMapBuilder* photonMapper = nullptr) {
        const int W = camera->imgWidth, H = camera->imgHeight;
        std::cout << "CastRays: image size W=" << W << " H=" << H << "\n";
        stopFlag = false;
        if (!sampling.devices.empty() && sampling.devices != g_rtDevices) {
            ReleaseGpuScenes();   // the device copies are rebuilt on the new device list
            g_rtDevices = sampling.devices;
        }
        GpuScene& gs = GetGpuScene(scene);
        rtx_camera cam{};
        cam.pos[0] = camera->pos.x; cam.pos[1] = camera->pos.y; cam.pos[2] = camera->pos.z;
        cam.dir[0] = camera->dir.x; cam.dir[1] = camera->dir.y; cam.dir[2] = camera->dir.z;
        cam.up[0] = camera->up.x; cam.up[1] = camera->up.y; cam.up[2] = camera->up.z;
        cam.fov = camera->fov;
        cam.focaldist = camera->focaldist;
        cam.dof = camera->dof;
        cam.width = W;
        cam.height = H;
        cam.srgb = camera->sRGB ? 1 : 0;
        rtx_render_params rp{};
        rp.spp_min = sampling.minSamples;
        rp.spp_max = progressivePreview ? sampling.minSamples : sampling.maxSamples;
        rp.index_stride = 1024;
        rp.ci_halfwidth = sampling.maxHalfWidth;
        rp.filter_type = (int)g_filterType;
        rp.filter_radius = g_filterRadius;
        rp.max_bounces = sampling.maxBounces;
        rp.seed = sampling.seed;
        rp.tile_size = 16;
        rp.rank = sampling.rank;
        rp.world = sampling.world;
        rp.samples_per_pass = sampling.samplesPerPass;
        rp.shading_mode = sampling.legacyShading ? RTX_SHADING_LEGACY : RTX_SHADING_PATH;
        rtx_photons* ph = photonMapper ? photonMapper->Handle() : nullptr;
        rp.use_photons = ph ? 1 : 0;
        active = gs.Handle();
        activeGroup = gs.Group();
        if (gs.Group() && !progressivePreview) {
            // the frame's buckets dealt to the GPUs of the group, films merged on the first one (rtx_group_render)
            rp.use_photons = rtx_group_photons(gs.Group(), 0) && photonMapper ? 1 : 0;
            int rc = rtx_group_render(gs.Group(), &cam, &rp, &image.GetLinearPixels()->r, &image.GetPixels()->r, image.GetZBuffer(),
                                      image.GetSampleCount(), &image.GetAlbedo()->r, &image.GetNormal()->x, &lastStats);
            active = nullptr;
            activeGroup = nullptr;
            if (rc != RTX_OK) {
                std::cerr << "CastRays failed: " << rtx_last_error() << "\n";
                return;
            }
            image.IncrementNumRenderPixel(W * H);
            return;
        }
        if (progressivePreview) {
            CastRaysProgressive(image, gs.Handle(), cam, rp, ph);
            active = nullptr;
            return;
        }
        static_assert(sizeof(Color) == 12 && sizeof(Color24) == 3, "RenderImage arrays must be tightly packed");
        int rc = rtx_render(gs.Handle(), &cam, &rp, ph, &image.GetLinearPixels()->r, &image.GetPixels()->r, image.GetZBuffer(),
                            image.GetSampleCount(), &lastStats);
        active = nullptr;
        if (rc != RTX_OK) {
            std::cerr << "CastRays failed: " << rtx_last_error() << "\n";   // logged; the caller keeps running (§8b)
            return;
        }
        image.IncrementNumRenderPixel(W * H);
    }
    // Progressive mode (RayCaster.cpp:1025-1144, 1215-1279): batches of one eighth of the buckets take
    // kProgressiveSamplesPerPass samples per pixel per visit, the image is refreshed after every batch, and the loop
    // ends when every pixel holds maxSamples (no confidence-interval test) or Stop() is called.
    void CastRaysProgressive(RenderImage& image, rtx_scene* gpu, const rtx_camera& cam, const rtx_render_params& rp, rtx_photons* ph) {
        const int W = cam.width, H = cam.height, bs = 16;
        rtx_film film{};
        rtx_pixel_stats pst{};
        if (rtx_film_create(W, H, &film, &pst) != RTX_OK) {
            std::cerr << "CastRays failed: " << rtx_last_error() << "\n";
            return;
        }
        std::vector<Bucket> buckets;   // RayCaster.cpp:1152-1158
        for (int y = 0; y < H; y += bs)
            for (int x = 0; x < W; x += bs) buckets.push_back(Bucket{x, y, std::min(x + bs, W), std::min(y + bs, H)});
        std::vector<int> taken(buckets.size(), 0);
        const size_t batchSize = std::max<size_t>(1, buckets.size() / 8);
        const int perPass = std::max(1, sampling.progressiveSamplesPerPass);
        size_t cursor = 0;
        bool allDone = false, seeded = false;
        image.ResetNumRenderedPixels();
        std::vector<int> pixels;
        while (!stopFlag && !allDone) {
            const size_t batchEnd = std::min(cursor + batchSize, buckets.size());
            // buckets of a batch that hold the same number of samples go into one round
            for (size_t b = cursor; b < batchEnd && !stopFlag;) {
                const int base = taken[b];
                pixels.clear();
                size_t e = b;
                for (; e < batchEnd && taken[e] == base; ++e)
                    for (int y = buckets[e].y0; y < buckets[e].y1; ++y)
                        for (int x = buckets[e].x0; x < buckets[e].x1; ++x) pixels.push_back(y * W + x);
                const int n = std::min(perPass, sampling.maxSamples - base);
                if (n > 0) {
                    if (rtx_render_round(gpu, &cam, &rp, ph, &film, &pst, pixels.data(), pixels.size(), base, n, 1, &lastStats) != RTX_OK) {
                        std::cerr << "CastRays failed: " << rtx_last_error() << "\n";
                        rtx_film_destroy(&film, &pst);
                        return;
                    }
                    for (size_t k = b; k < e; ++k) taken[k] = base + n;
                }
                b = e;
            }
            // mergeToImage
            if (rtx_film_resolve(&film, &cam, &image.GetLinearPixels()->r, &image.GetPixels()->r, image.GetZBuffer(),
                                 image.GetSampleCount(), nullptr, nullptr) != RTX_OK)
                std::cerr << "CastRays failed: " << rtx_last_error() << "\n";
            image.IncrementNumRenderPixel(seeded ? 1 : W * H);
            seeded = true;
            if (stopFlag) break;
            allDone = true;
            for (int t : taken)
                if (t < sampling.maxSamples) {
                    allDone = false;
                    break;
                }
            cursor = batchEnd >= buckets.size() ? 0 : batchEnd;
        }
        rtx_film_destroy(&film, &pst);
    }
    void Stop() {
        stopFlag = true;
        if (rtx_group* g = activeGroup.load()) rtx_group_cancel(g);
        else if (rtx_scene* s = active.load()) rtx_cancel(s);
    }
    bool IsStopped() const { return stopFlag.load(); }
    void SetProgressivePreview(bool progressive) { progressivePreview = progressive; }

    SamplingConfig sampling;
    rtx_stats lastStats{};

private:
    Scene* scene;
    Camera* camera;
    std::atomic<bool> stopFlag{false};
    std::atomic<rtx_scene*> active{nullptr};
    std::atomic<rtx_group*> activeGroup{nullptr};
    bool progressivePreview = false;
};

// RayCaster.h:82-83, RayCaster.cpp:1406-1422 (free-function forms; RayCastRenderer below is the reference's class)
inline bool TraceRay(const Scene& scene, Ray const& ray, HitInfo& hInfo, int hitSide = HIT_FRONT_AND_BACK) {
    hInfo.Init();
    ExtendedHitInfo h;
    h.Init();
    GpuScene& gs = GetGpuScene(&scene);
    rtx_ray r = detail::ToRay(ray, h.z);
    rtx_hit out;
    if (rtx_trace_closest(gs.Handle(), &r, 1, hitSide, 0, &out) != RTX_OK || out.obj < 0) return false;
    detail::FillHit(gs, out, hInfo);
    return true;
}
// Declared bool in the reference but returns 0.0f / 1.0f (RayCaster.cpp:1412-1422): false = occluded before t_max
inline bool TraceShadowRay(const Scene& scene, Ray const& ray, float t_max, int hitSide = HIT_FRONT_AND_BACK) {
    HitInfo h;
    if (TraceRay(scene, ray, h, hitSide)) {
        float d = (h.p - ray.p).Length();
        if (d < t_max) return false;
    }
    return true;
}

// Stand-in for the framework's Renderer base as far as RayCastRenderer uses it (renderer.h is not in the reference
// repository): owns the Scene, the Camera and the RenderImage; LoadScene goes through the host's scene loader.
class Renderer {
public:
    virtual ~Renderer() = default;
    Scene& GetScene() { return scene; }
    Scene const& GetScene() const { return scene; }
    Camera& GetCamera() { return camera; }
    RenderImage& GetRenderImage() { return renderImage; }
    bool IsRendering() const { return isRendering; }
    // framework Renderer::LoadScene (xmlload.cpp, absent): the host's loader for the cerealBar.xml dialect (SceneLoader.h)
    virtual bool LoadScene(char const* sceneFilename) {
        LoadedScene ls;
        SceneLoader loader;
        loader.allowMissingMeshes = true;
        std::string err;
        if (!loader.Load(sceneFilename, ls, err)) {
            std::cerr << "LoadScene failed: " << err << "\n";
            return false;
        }
        ReleaseGpuScenes();            // device copies of the scene that is being replaced
        scene = std::move(ls.scene);
        camera = ls.camera;
        renderImage.Init(camera.imgWidth, camera.imgHeight);
        return true;
    }
    virtual void BeginRender() = 0;
    virtual void StopRender() = 0;
    virtual bool TraceRay(Ray const& ray, HitInfo& hInfo, int hitSide = HIT_FRONT_AND_BACK) const = 0;
    virtual bool TraceShadowRay(Ray const& ray, float t_max, int hitSide = HIT_FRONT_AND_BACK) const = 0;
    virtual PhotonMap const* GetPhotonMap() const { return nullptr; }
    virtual PhotonMap const* GetCausticsMap() const { return nullptr; }

protected:
    Scene scene;
    Camera camera;
    RenderImage renderImage;
    std::atomic<bool> isRendering{false};
};

// RayCaster.h:66-117 / RayCaster.cpp:1321-1430: the renderer that runs RayCaster::CastRays on a background thread and
// answers the framework's single-ray queries. Same members and semantics; the denoise hook is the host's (Denoiser.h).
class RayCastRenderer : public Renderer {
public:
    RayCastRenderer() = default;
    ~RayCastRenderer() override { StopRender(); }

    bool LoadScene(char const* sceneFilename) override {
        if (!Renderer::LoadScene(sceneFilename)) return false;
        raycaster = std::make_unique<RayCaster>(&GetScene(), &GetCamera());
        raycaster->SetProgressivePreview(progressivePreview);
        return true;
    }
    // a scene built in memory instead of loaded from XML (the stand-in Renderer owns it: fill GetScene()/GetCamera() first)
    void AttachScene() {
        raycaster = std::make_unique<RayCaster>(&GetScene(), &GetCamera());
        raycaster->SetProgressivePreview(progressivePreview);
    }
    RayCaster* GetRayCaster() { return raycaster.get(); }

    // RayCaster.cpp:1321-1398
    void BeginRender() override {
        if (!raycaster) return;
        if (renderThread.joinable()) renderThread.join();
        isRendering = true;
        renderThread = std::thread([this]() {
            try {
                const auto t0 = std::chrono::steady_clock::now();
                raycaster->CastRays(this->GetRenderImage(), photonMapper);
                const double total_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                const int hours = (int)total_seconds / 3600, minutes = ((int)total_seconds % 3600) / 60, seconds = (int)total_seconds % 60;
                char buf[32];
                std::snprintf(buf, sizeof(buf), "%02d:%02d:%02d", hours, minutes, seconds);
                std::cout << "Render time: " << buf << "\n";
                if (!isPreviewMode) {
                    if (autoSaveOutputs) {
                        this->GetRenderImage().ComputeZBufferImage();
                        this->GetRenderImage().ComputeSampleCountImage();
                        this->GetRenderImage().SaveImage(outputColorPath.c_str());
                        this->GetRenderImage().SaveZImage(outputZPath.c_str());
                        this->GetRenderImage().SaveSampleCountImage(outputSamplesPath.c_str());
                    }
                    isRendering = false;
                }
                // preview mode keeps isRendering set so that the viewport stays open (RayCaster.cpp:1385)
            } catch (const std::exception& e) {
                std::cerr << "Render thread exception: " << e.what() << "\n";
                isRendering = false;
            } catch (...) {
                std::cerr << "Render thread unknown exception.\n";
                isRendering = false;
            }
        });
    }
    // RayCaster.cpp:1400-1404
    void StopRender() override {
        if (raycaster) raycaster->Stop();
        if (renderThread.joinable()) renderThread.join();
        isRendering = false;
    }
    // blocks until the frame started by BeginRender is done (Main.cpp:354-358 polls IsRendering every 50 ms instead)
    void WaitRender() {
        if (renderThread.joinable()) renderThread.join();
    }

    // RayCaster.cpp:1406-1410
    bool TraceRay(Ray const& ray, HitInfo& hInfo, int hitSide = HIT_FRONT_AND_BACK) const override {
        hInfo.Init();
        return IntersectNode(&GetScene().rootNode, ray, hInfo, hitSide);
    }
    // RayCaster.cpp:1412-1422 (declared bool, returns 0.0f / 1.0f: false = occluded before t_max)
    bool TraceShadowRay(Ray const& ray, float t_max, int hitSide = HIT_FRONT_AND_BACK) const override {
        HitInfo h;
        h.Init();
        if (IntersectNode(&GetScene().rootNode, ray, h, hitSide)) {
            const float d = (h.p - ray.p).Length();
            if (d < t_max) return 0.0f;
        }
        return 1.0f;
    }
    PhotonMap const* GetPhotonMap() const override { return photonMapper ? photonMapper->GetPhotonMap() : nullptr; }       // :1424-1426
    PhotonMap const* GetCausticsMap() const override { return photonMapper ? photonMapper->GetCausticsMap() : nullptr; }   // :1428-1430

    void SetPreviewMode(bool preview) { isPreviewMode = preview; }
    void SetProgressivePreview(bool progressive) {
        progressivePreview = progressive;
        if (raycaster) raycaster->SetProgressivePreview(progressive);
    }
    void SetOutputPaths(const std::string& color, const std::string& z, const std::string& samples, const std::string& denoised) {
        outputColorPath = color;
        outputZPath = z;
        outputSamplesPath = samples;
        outputDenoisedPath = denoised;
    }
    void SetAutoSaveOutputs(bool enable) { autoSaveOutputs = enable; }
    void SetPhotonMapper(PhotonMapBuilder* mapper) { photonMapper = mapper; }

private:
    std::unique_ptr<RayCaster> raycaster;
    std::thread renderThread;
    bool isPreviewMode = false;
    bool progressivePreview = false;
    PhotonMapBuilder* photonMapper = nullptr;
    std::string outputColorPath = "render.png";
    std::string outputZPath = "zbuffer.png";
    std::string outputSamplesPath = "samples.png";
    std::string outputDenoisedPath = "render_denoised.png";
    bool autoSaveOutputs = true;
};

}  // namespace rtamd
