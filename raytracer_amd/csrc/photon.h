// Photon maps in HBM behind the rtx_photons handle (host side). Replaces PhotonMapBuilder's two cy::PhotonMap
// members (reference RayTracer/PhotonMapBuilder.h:42-43).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pathstate.h"

// Raw stored photon, 48 B, the unit that is all-gathered between GPUs:
//   {pos.xyz, bits(key)}, {dir.xyz, 0}, {power.rgb (unscaled), 0};  key = photonId * 4 + bounce
struct DevPhotonRec {
    float4 pos, dir, power;
};

struct PhotonSet {
    DevPhotonRec* raw = nullptr;   // capacity records
    uint32_t capacity = 0;
    uint32_t n = 0;
    // sorted-by-cell copy used by the gather
    float4* pos = nullptr;
    float4* dir = nullptr;
    float4* power = nullptr;
    uint32_t* cellStart = nullptr;
    size_t cells = 0;
    int gx = 0, gy = 0, gz = 0;
    float cell = 0.0f;
    int imin[3] = {0, 0, 0};       // first lattice cell of the grid window on each axis
    float radius = 0.0f;
};

struct rtx_photons {
    struct rtx_scene* scene = nullptr;
    PhotonSet set[2];              // 0 global, 1 caustic
    int searchCount = 100;
    int numPhotons = 0;
    uint32_t* counts = nullptr;    // device: [0] global stored, [1] caustic stored, [2] work cursor, [3] segments traced
};

void photon_device_map(const rtx_photons* p, int which, DevPhotonMap& out);
