// Host-callable launch wrappers of the HIP kernels (definitions in kernels_trace.hip / kernels_path.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "layout.h"
#include "../../include/rtx.h"

// Camera with the per-frame constants of RayCaster.cpp:717-724 / CameraAdapter.h:8-21 evaluated once on the host.
struct DevCamera {
    float pos[3], f[3], r[3], u[3];
    float halfX, halfY;
    float focus, aperture;
    int W, H;
    int thinLens;       // camera->dof > 1e-6f
    int srgb;
    float fov;
};

struct LaunchCfg {
    int gridBlocks;          // persistent grid: CUs x resident blocks
    uint2* spill;            // gridBlocks*256*STACK_GLOBAL entries
    uint32_t* workCounter;   // zeroed before every launch that fetches work dynamically
    unsigned long long* counters;  // [0] nodes, [1] tris, [2] rays (only with counting)
    hipStream_t stream;
    int counting;
};

// extend: closest hit for n rays (rtx_ray layout, 2 x float4) -> hits float4 {t,u,v,bits(slot)}
void launch_extend(const DevScene& sc, const LaunchCfg& cfg, const float4* rays, uint32_t n, int hitSide, int withLights,
                   float4* hits);
// CameraAdapter.h:7-27 pixel-centre rays for the whole image
void launch_primary_raygen(const DevCamera& cam, float4* rays, hipStream_t stream);
// compact hit -> full rtx_hit (device array)
void launch_expand_hits(const DevScene& sc, const float4* rays, const float4* hits, uint32_t n, rtx_hit* out,
                        hipStream_t stream);
// Visible()/VisibleSimple(): rays = segments (o, unit d, tmax = distance), normals float4 per ray
void launch_visibility(const DevScene& sc, const LaunchCfg& cfg, const float4* rays, const float4* normals, uint32_t n,
                       int useTransmittance, float* vis, unsigned long long* hopCounter);

int query_resident_blocks(int which);  // blocks per CU for kernel 'which' (0 extend, 1 shadow)
