// Mirror of the reference's RayTracer/PhotonMapBuilder.h: same class, setters/getters and BuildPhotonMap(Scene*);
// emission, tracing and the map itself live on the GPU (rtx_photon_build, include/rtx.h).
#pragma once
#include <iostream>
#include "SceneAdapter.h"

namespace rtamd {

// Stand-in for cy::PhotonMap as seen by callers: a view of one of the two device-resident maps.
class PhotonMap {
public:
    int NumPhotons() const {
        uint64_t c[2] = {0, 0};
        if (!owner) return 0;
        rtx_photon_info(owner, c, nullptr);
        return (int)c[which];
    }
    // PhotonMap::EstimateIrradiance<N, PHOTONMAP_FILTER_LINEAR> for one point (RayCaster.cpp:143-153)
    void EstimateIrradiance(Color& irrad, Vec3f const& pos, Vec3f const& normal) const {
        float out[3] = {0, 0, 0};
        if (owner) rtx_photon_estimate(owner, which, &pos.x, &normal.x, 1, out);
        irrad = Color(out[0], out[1], out[2]);
    }
    rtx_photons* owner = nullptr;
    int which = 0;
};

class PhotonMapBuilder {
public:
    PhotonMapBuilder() = default;
    virtual ~PhotonMapBuilder() { Release(); }

    // PhotonMapBuilder.cpp:20-60
    void BuildPhotonMap(Scene* scene) {
        if (!scene) return;
        Release();
        photonMap.owner = causticsMap.owner = nullptr;
        rtx_photon_params p{};
        p.num_photons = numPhotons;
        p.max_bounces = maxBounces;
        p.photon_radius = photonRadiusOverridden ? photonSearchRadius : 0.0f;     // AdjustPhotonGatherSettings :70-77
        p.caustic_radius = causticRadiusOverridden ? causticSearchRadius : 0.0f;
        p.search_count = photonSearchCount;
        p.seed = seed;
        p.first_photon = 0;
        p.photon_stride = 1;
        std::cout << "Building photon map with " << numPhotons << " photons...\n";
        GpuScene& gs = GetGpuScene(scene);
        if (gs.Group()) {
            // several GPUs: every member traces its share of the photon ids, all of them end up holding the whole map
            // (the reference's workers feed one map, :102-146); the group owns the maps, member 0's is the one queried here
            if (rtx_group_photon_build(gs.Group(), &p, nullptr) != RTX_OK) {
                std::cerr << "Photon map build failed: " << rtx_last_error() << "\n";
                return;
            }
            handle = rtx_group_photons(gs.Group(), 0);
            borrowed = true;
            borrowedGeneration = GpuSceneGeneration();
        } else if (rtx_photon_build(gs.Handle(), &p, &handle, nullptr) != RTX_OK) {
            std::cerr << "Photon map build failed: " << rtx_last_error() << "\n";   // logged, render continues (:54-59)
            return;
        }
        float radii[2];
        uint64_t counts[2];
        rtx_photon_info(handle, counts, radii);
        photonSearchRadius = radii[0];
        causticSearchRadius = radii[1];
        photonMap.owner = causticsMap.owner = handle;
        photonMap.which = 0;
        causticsMap.which = 1;
        std::cout << "Photon map contains " << counts[0] << " photons\nCaustics map contains " << counts[1] << " photons\n";
    }

    PhotonMap const* GetPhotonMap() const { Validate(); return &photonMap; }
    PhotonMap const* GetCausticsMap() const { Validate(); return &causticsMap; }
    // nullptr once the GPU group that owns the maps is gone (ReleaseGpuScenes: a new device list, a new scene): the maps
    // went with it, and the next BuildPhotonMap makes new ones
    rtx_photons* Handle() const { Validate(); return handle; }

    // PhotonMapBuilder.h:26-39
    void SetNumPhotons(int n) { numPhotons = n; }
    void SetPhotonSearchRadius(float r) {
        photonSearchRadius = r;
        photonRadiusOverridden = true;
        if (!causticRadiusOverridden) {
            causticSearchRadius = r * 0.25f;
            if (causticSearchRadius < 0.01f) causticSearchRadius = 0.01f;
        }
    }
    void SetCausticSearchRadius(float r) { causticSearchRadius = r; causticRadiusOverridden = true; }
    void SetPhotonSearchCount(int c) { photonSearchCount = c; }
    float GetPhotonSearchRadius() const { return photonSearchRadius; }
    float GetCausticSearchRadius() const { return causticSearchRadius; }
    int GetPhotonSearchCount() const { return photonSearchCount; }
    void SetSeed(uint64_t s) { seed = s; }

private:
    void Validate() const {
        if (handle && borrowed && borrowedGeneration != GpuSceneGeneration()) {
            handle = nullptr;
            borrowed = false;
            photonMap.owner = causticsMap.owner = nullptr;
        }
    }
    void Release() {
        Validate();
        if (handle && !borrowed) rtx_photons_destroy(handle);
        handle = nullptr;
        borrowed = false;
    }
    mutable PhotonMap photonMap, causticsMap;
    mutable rtx_photons* handle = nullptr;
    mutable bool borrowed = false;      // the handle belongs to a GPU group (SceneAdapter.h: g_rtDevices) ...
    unsigned long borrowedGeneration = 0;   // ... and is valid while no ReleaseGpuScenes() has run since
    int numPhotons = 10000000;          // PhotonMapBuilder.h:46-54
    float photonSearchRadius = 0.2f;
    float causticSearchRadius = 0.05f;
    int photonSearchCount = 100;
    bool photonRadiusOverridden = false;
    bool causticRadiusOverridden = false;
    int maxBounces = 2;
    uint64_t seed = 0x9e3779b97f4a7c15ULL;
};

}  // namespace rtamd
