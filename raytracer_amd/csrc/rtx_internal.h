// Internal host-side state behind the opaque C-ABI handles.
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <atomic>
#include <string>
#include <vector>
#include "bvh_build.h"
#include "kernels.h"

void rtx_set_error(const std::string& msg);
int rtx_fail(int code, const std::string& msg);
#define RTX_HIP(call)                                                                               \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return rtx_fail(RTX_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));        \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need);   // grows (never shrinks); returns RTX_OK or error
    void release();
};

struct rtx_scene {
    int device = 0;
    DevScene dev{};
    std::vector<void*> owned;      // device allocations released on destroy
    size_t deviceBytes = 0;
    uint64_t nTris = 0, nNodes = 0;
    int depth = 0;
    LaunchCfg cfg{};
    int numCUs = 0;
    int hasMicrofacet = 0;        // some material is RTX_MTL_MICROFACET: selects the two-BSDF shade kernel
    int hasPhong = 0;             // some material is RTX_MTL_PHONG or textures Specular / Reflection / Refraction / Glossiness: selects the
                                  // shade kernel that carries the Phong form of the lobes and those texture fetches
    std::atomic<int> cancel{0};   // rtx_cancel
    // reusable scratch for the host-array entry points
    DevBuf rays, hits, full, aux, aux2;
    // path-tracing pool (kernels_path.hip)
    struct PathPool* pool = nullptr;
};

int make_dev_camera(const rtx_camera* cam, DevCamera& out);

// The scene descriptor a traversal kernel is launched with outside the wavefront path loop (ray queries, photon emission,
// the legacy integrator): with RTX_AMD_BOX_CULLS=1 a scene of world-space triangles runs the general instantiation there,
// which repeats the reference's root and object box tests (traverse.h ref_box_hit). The path loop
// itself keeps its lean kernels and checks after the fact (rtx_render.cpp).
inline DevScene rtx_traversal_scene(const rtx_scene* s) {
    DevScene d = s->dev;
    if (!d.has_xf)
        if (const char* e = std::getenv("RTX_AMD_BOX_CULLS"))
            if (std::atoi(e) != 0) d.has_xf = 1;
    return d;
}
