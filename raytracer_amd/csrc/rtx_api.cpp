// C ABI (include/rtx.h): device management, scene upload and the ray-casting entry points.
#include "rtx_internal.h"
#include "rtx_math.h"
#include <cmath>
#include <cstdlib>
#include <cstring>

static thread_local std::string g_err;
void rtx_set_error(const std::string& msg) { g_err = msg; }
int rtx_fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

int DevBuf::ensure(size_t need) {
    if (need <= bytes) return RTX_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
    size_t cap = need + need / 8 + 256;
    hipError_t e = hipMalloc(&p, cap);
    if (e != hipSuccess) return rtx_fail(RTX_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
    bytes = cap;
    return RTX_OK;
}
void DevBuf::release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
}

extern "C" {

const char* rtx_last_error(void) { return g_err.c_str(); }
const char* rtx_version(void) { return "raytracer_amd 0.1 (gfx950)"; }

int rtx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int rtx_set_device(int device) {
    if (rtx_device_count() <= 0) return rtx_fail(RTX_ERR_NO_DEVICE, "no HIP device present");
    RTX_HIP(hipSetDevice(device));
    return RTX_OK;
}

}  // extern "C"

template <class T>
static int upload(rtx_scene* s, const std::vector<T>& v, const T** out) {
    *out = nullptr;
    if (v.empty()) return RTX_OK;
    void* p = nullptr;
    RTX_HIP(hipMalloc(&p, v.size() * sizeof(T)));
    s->owned.push_back(p);
    s->deviceBytes += v.size() * sizeof(T);
    RTX_HIP(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = (const T*)p;
    return RTX_OK;
}

int make_dev_camera(const rtx_camera* cam, DevCamera& c) {
    if (!cam || cam->width <= 0 || cam->height <= 0) return rtx_fail(RTX_ERR_INVALID, "invalid camera");
    // RayCaster.cpp:717-724 (same constants as CameraAdapter.h:8-21), evaluated once per frame in fp32
    const float fovRad = cam->fov * 3.14159265f / 180.0f;
    c.halfY = rtx_tanf(0.5f * fovRad);   // the routine the kernels and the oracle use (rtx_math.h)
    const float aspect = (cam->height > 0) ? float(cam->width) / float(cam->height) : 1.0f;
    c.halfX = c.halfY * aspect;
    const float* f = cam->dir;
    const float* up = cam->up;
    float r[3] = {f[1] * up[2] - f[2] * up[1], f[2] * up[0] - f[0] * up[2], f[0] * up[1] - f[1] * up[0]};
    float rl = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    for (int k = 0; k < 3; ++k) r[k] = r[k] / rl;
    float u[3] = {r[1] * f[2] - r[2] * f[1], r[2] * f[0] - r[0] * f[2], r[0] * f[1] - r[1] * f[0]};
    float ul = std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    for (int k = 0; k < 3; ++k) u[k] = u[k] / ul;
    for (int k = 0; k < 3; ++k) {
        c.pos[k] = cam->pos[k];
        c.f[k] = f[k];
        c.r[k] = r[k];
        c.u[k] = u[k];
    }
    c.focus = cam->focaldist;
    c.aperture = cam->dof;
    c.thinLens = cam->dof > 1e-6f ? 1 : 0;
    c.W = cam->width;
    c.H = cam->height;
    c.srgb = cam->srgb;
    c.fov = cam->fov;
    if (!(rl > 0.0f) || !(ul > 0.0f)) return rtx_fail(RTX_ERR_INVALID, "camera dir and up are parallel");
    return RTX_OK;
}

extern "C" {

void rtx_scene_destroy(rtx_scene* s);

int rtx_scene_create(const rtx_scene_desc* desc, rtx_scene** out) {
    if (!out) return rtx_fail(RTX_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!desc) return rtx_fail(RTX_ERR_INVALID, "desc is NULL");
    if (desc->n_lights > RTX_MAX_LIGHTS) return rtx_fail(RTX_ERR_INVALID, "more than RTX_MAX_LIGHTS lights");
    if (rtx_device_count() <= 0) return rtx_fail(RTX_ERR_NO_DEVICE, "no HIP device present (this library has no CPU path)");
    HostScene hs;
    std::string err;
    if (!build_host_scene(desc, hs, err)) return rtx_fail(RTX_ERR_INVALID, err);
    if (3 * (hs.depth + 2) > STACK_LDS + STACK_GLOBAL)
        return rtx_fail(RTX_ERR_TOO_LARGE, "BVH depth " + std::to_string(hs.depth) + " exceeds the traversal stack budget");

    rtx_scene* s = new rtx_scene();
    (void)hipGetDevice(&s->device);
    int rc;
#define CHECK_RC(x)               \
    if ((rc = (x)) != RTX_OK) {   \
        rtx_scene_destroy(s);     \
        return rc;                \
    }
    DevScene& d = s->dev;
    std::memset(&d, 0, sizeof(d));
    CHECK_RC(upload(s, hs.nodes, &d.nodes));
    CHECK_RC(upload(s, hs.tris, &d.tris));
    CHECK_RC(upload(s, hs.attrs, &d.attrs));
    CHECK_RC(upload(s, hs.objs, &d.objs));
    CHECK_RC(upload(s, hs.objXf, &d.objXf));
    d.root = hs.root;
    d.n_tris = (int)hs.tris.size();
    d.n_nodes = (int)hs.nodes.size();
    d.n_objs = (int)hs.objs.size();
    for (const ObjRec& o : hs.objs)
        if (o.flags & OBJ_XFMESH) d.has_xf = 1;   // selects the kernels that carry the object-space triangle test ...
    for (const TriRec& t : hs.tris)
        if (t.kind != 0u) d.has_xf = 1;           // ... and the Sphere / Plane test (any record that is not a world-space triangle)
    // RTX_AMD_GENERAL_KERNELS=1: the general kernels for a scene of world-space triangles too. They repeat the reference's
    // root and object box tests, which the lean kernels leave out: the switch measures what that omits
    if (const char* e = std::getenv("RTX_AMD_GENERAL_KERNELS")) if (std::atoi(e) != 0) d.has_xf = 1;
    std::memcpy(d.bounds, hs.bounds, sizeof(d.bounds));
    s->nTris = hs.tris.size();
    s->nNodes = hs.nodes.size();
    s->depth = hs.depth;

    // textures: RGB8 -> uchar4, RGB32F -> float4
    std::vector<DevTexMap> tms(desc->n_texmaps);
    for (int i = 0; i < desc->n_texmaps; ++i) {
        const rtx_texmap_desc& t = desc->texmaps[i];
        if (t.width <= 0 || t.height <= 0 || !t.texels) {
            rtx_scene_destroy(s);
            return rtx_fail(RTX_ERR_INVALID, "texture " + std::to_string(i) + " is empty");
        }
        const size_t n = (size_t)t.width * t.height;
        void* p = nullptr;
        if (t.format == RTX_TEX_RGB8) {
            std::vector<uint8_t> tmp(n * 4);
            const uint8_t* src = (const uint8_t*)t.texels;
            for (size_t k = 0; k < n; ++k) {
                tmp[4 * k] = src[3 * k];
                tmp[4 * k + 1] = src[3 * k + 1];
                tmp[4 * k + 2] = src[3 * k + 2];
                tmp[4 * k + 3] = 255;
            }
            if (hipMalloc(&p, n * 4) != hipSuccess || hipMemcpy(p, tmp.data(), n * 4, hipMemcpyHostToDevice) != hipSuccess) {
                rtx_scene_destroy(s);
                return rtx_fail(RTX_ERR_HIP, "texture upload failed");
            }
            s->deviceBytes += n * 4;
        } else if (t.format == RTX_TEX_RGB32F) {
            std::vector<float> tmp(n * 4);
            const float* src = (const float*)t.texels;
            for (size_t k = 0; k < n; ++k) {
                tmp[4 * k] = src[3 * k];
                tmp[4 * k + 1] = src[3 * k + 1];
                tmp[4 * k + 2] = src[3 * k + 2];
                tmp[4 * k + 3] = 1.0f;
            }
            if (hipMalloc(&p, n * 16) != hipSuccess || hipMemcpy(p, tmp.data(), n * 16, hipMemcpyHostToDevice) != hipSuccess) {
                rtx_scene_destroy(s);
                return rtx_fail(RTX_ERR_HIP, "texture upload failed");
            }
            s->deviceBytes += n * 16;
        } else {
            rtx_scene_destroy(s);
            return rtx_fail(RTX_ERR_INVALID, "unknown texture format");
        }
        s->owned.push_back(p);
        tms[i].texels = p;
        tms[i].format = t.format;
        tms[i].width = t.width;
        tms[i].height = t.height;
        tms[i].pad = 0;
        std::memcpy(tms[i].xf, t.xform, sizeof(float) * 12);
    }
    CHECK_RC(upload(s, tms, &d.texmaps));

    auto texOK = [&](int t) { return t >= -1 && t < desc->n_texmaps; };
    std::vector<DevMaterial> mtls(desc->n_materials);
    for (int i = 0; i < desc->n_materials; ++i) {
        const rtx_material_desc& m = desc->materials[i];
        if (!texOK(m.diffuse_tex) || !texOK(m.specular_tex) || !texOK(m.reflection_tex) || !texOK(m.refraction_tex) ||
            !texOK(m.emission_tex) || !texOK(m.bump_tex) || !texOK(m.glossiness_tex) || (m.type != RTX_MTL_BLINN && m.type != RTX_MTL_PHONG && m.type != RTX_MTL_MICROFACET)) {
            rtx_scene_destroy(s);
            return rtx_fail(RTX_ERR_INVALID, "material " + std::to_string(i) + ": bad texture index or type");
        }
        DevMaterial& o = mtls[i];
        o.type = m.type;
        if (m.type == RTX_MTL_MICROFACET) s->hasMicrofacet = 1;
        if (m.type == RTX_MTL_PHONG) s->hasPhong = 1;
        // ... or a texture on Specular / Reflection / Refraction / Glossiness (a MtlMicrofacet reads its own maps in build_micro_ctx)
        if (m.type != RTX_MTL_MICROFACET && (m.specular_tex >= 0 || m.reflection_tex >= 0 || m.refraction_tex >= 0 || m.glossiness_tex >= 0)) s->hasPhong = 1;
        std::memcpy(o.kd, m.diffuse, 12);    o.kd_tex = m.diffuse_tex;
        std::memcpy(o.ks, m.specular, 12);   o.ks_tex = m.specular_tex;
        std::memcpy(o.kr, m.reflection, 12); o.kr_tex = m.reflection_tex;
        std::memcpy(o.kt, m.refraction, 12); o.kt_tex = m.refraction_tex;
        std::memcpy(o.ke, m.emission, 12);   o.ke_tex = m.emission_tex;
        o.gloss = m.glossiness;
        o.gloss_tex = m.glossiness_tex;
        o.ior = m.ior;
        std::memcpy(o.absorption, m.absorption, 12);
        o.bump_tex = m.bump_tex;
        o.bump_scale = m.bump_scale;
    }
    CHECK_RC(upload(s, mtls, &d.mtls));
    d.n_mtls = desc->n_materials;
    d.n_texmaps = desc->n_texmaps;
    d.n_lights = desc->n_lights;
    for (int i = 0; i < desc->n_lights; ++i) {
        std::memcpy(d.lights[i].pos, desc->lights[i].position, 12);
        std::memcpy(d.lights[i].intensity, desc->lights[i].intensity, 12);
        d.lights[i].size = desc->lights[i].size;
        d.lights[i].attenuation = desc->lights[i].attenuation;
        DevLight& L = d.lights[i];
        if (L.size > 0.0f) {   // (this file is built with -ffp-contract=off like the kernels: the same bits as the expressions in bsdf.h)
            const float PI_F = 3.14159265358979323846f;
            const float area = 4.0f * PI_F * L.size * L.size;
            L.pdfArea = 2.0f / area;
            const float att_factor = (L.attenuation == 0.0f) ? 1.0f : L.attenuation * L.attenuation;
            for (int k = 0; k < 3; ++k) {
                L.radiance[k] = L.intensity[k] / (PI_F * L.size * L.size);
                L.radAtt[k] = L.radiance[k] / att_factor;
            }
        }
    }
    if (!texOK(desc->environment_tex) || !texOK(desc->background_tex)) {
        rtx_scene_destroy(s);
        return rtx_fail(RTX_ERR_INVALID, "environment/background texture index out of range");
    }
    std::memcpy(d.env, desc->environment, 12);
    d.env_tex = desc->environment_tex;
    std::memcpy(d.bg, desc->background, 12);
    d.bg_tex = desc->background_tex;
    d.bg_emission = desc->background_emission;

    // persistent launch geometry: CUs x resident blocks, spill slab for the deep part of the traversal stack
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, s->device) != hipSuccess) {
        rtx_scene_destroy(s);
        return rtx_fail(RTX_ERR_HIP, "hipGetDeviceProperties failed");
    }
    s->numCUs = prop.multiProcessorCount;
    int perCU = std::min(query_resident_blocks(0), 8);
    s->cfg.gridBlocks = s->numCUs * perCU;
    s->cfg.counting = 0;
    s->cfg.stream = nullptr;
    size_t spillBytes = (size_t)s->cfg.gridBlocks * TRAV_BLOCK * STACK_GLOBAL * sizeof(uint2);
    void* p = nullptr;
    if (hipMalloc(&p, spillBytes) != hipSuccess) {
        rtx_scene_destroy(s);
        return rtx_fail(RTX_ERR_HIP, "spill slab allocation failed");
    }
    s->owned.push_back(p);
    s->cfg.spill = (uint2*)p;
    s->deviceBytes += spillBytes;
    if (hipMalloc(&p, 256) != hipSuccess) {
        rtx_scene_destroy(s);
        return rtx_fail(RTX_ERR_HIP, "counter allocation failed");
    }
    s->owned.push_back(p);
    (void)hipMemset(p, 0, 256);
    s->cfg.workCounter = (uint32_t*)p;
    s->cfg.counters = (unsigned long long*)((char*)p + 64);
    if (hipMalloc(&p, SEGQ_WORDS * sizeof(uint32_t)) != hipSuccess) {
        rtx_scene_destroy(s);
        return rtx_fail(RTX_ERR_HIP, "queue control allocation failed");
    }
    s->owned.push_back(p);
    s->cfg.identQ.ctl = (uint32_t*)p;
#undef CHECK_RC
    RTX_HIP(hipDeviceSynchronize());
    *out = s;
    return RTX_OK;
}

void rtx_scene_release_pool(rtx_scene* s);

void rtx_scene_destroy(rtx_scene* s) {
    if (!s) return;
    rtx_scene_release_pool(s);
    for (void* p : s->owned) (void)hipFree(p);
    s->rays.release();
    s->hits.release();
    s->full.release();
    s->aux.release();
    s->aux2.release();
    delete s;
}

// Host-only checks of the flattening logic (no device needed): builds the BVH for desc and validates it.
// out[0]=triangles, [1]=nodes, [2]=depth. Returns RTX_OK or RTX_ERR_INVALID with the reason in rtx_last_error().
int rtx_debug_validate_bvh(const rtx_scene_desc* desc, uint64_t out[3]) {
    HostScene hs;
    std::string err;
    if (!build_host_scene(desc, hs, err)) return rtx_fail(RTX_ERR_INVALID, err);
    err = validate_host_scene(hs);
    if (!err.empty()) return rtx_fail(RTX_ERR_INVALID, err);
    if (out) {
        out[0] = hs.tris.size();
        out[1] = hs.nodes.size();
        out[2] = (uint64_t)hs.depth;
    }
    return RTX_OK;
}

// Host model of the traversal (bvh_build.cpp host_walk_closest) over n rays: closest hits and the number of nodes visited /
// triangles tested. Builds the tree for desc on the host; no device needed, nothing of the render path involved.
int rtx_debug_host_walk(const rtx_scene_desc* desc, const rtx_ray* rays, size_t n, float* t_out, int* prim_out, uint64_t counts[4]) {
    if (!desc || (n && (!rays || !t_out || !prim_out)) || !counts) return rtx_fail(RTX_ERR_INVALID, "bad argument");
    HostScene hs;
    std::string err;
    if (!build_host_scene(desc, hs, err)) return rtx_fail(RTX_ERR_INVALID, err);
    counts[0] = counts[1] = 0;
    counts[2] = hs.nodes.size();
    counts[3] = (uint64_t)hs.depth;
    for (size_t i = 0; i < n; ++i) host_walk_closest(hs, reinterpret_cast<const float*>(rays + i), t_out + i, prim_out + i, counts);
    return RTX_OK;
}

int rtx_debug_queue_layout_run(uint32_t producer_blocks, uint32_t block_threads, uint32_t mult, uint32_t run, uint32_t start[64],
                               uint32_t* segment_of_block) {
    if (!producer_blocks || !block_threads || !mult || !start) return rtx_fail(RTX_ERR_INVALID, "bad argument");
    for (uint32_t s = 0; s < QSEG; ++s) start[s] = segq_first_block(s, producer_blocks, run) * block_threads * mult;
    if (segment_of_block)
        for (uint32_t b = 0; b < producer_blocks; ++b) segment_of_block[b] = segq_segment_of_block(b, producer_blocks, run);
    return RTX_OK;
}
int rtx_debug_queue_layout(uint32_t producer_blocks, uint32_t block_threads, uint32_t mult, uint32_t start[64],
                           uint32_t* segment_of_block) {
    return rtx_debug_queue_layout_run(producer_blocks, block_threads, mult, 0u, start, segment_of_block);
}

int rtx_debug_math(int fn, const float* x, const float* y, size_t n, float* out, int on_device) {
    if (fn < 0 || fn > 8 || (n && (!x || !y || !out))) return rtx_fail(RTX_ERR_INVALID, "bad argument");
    if (n > 0x7FFFFFFFull) return rtx_fail(RTX_ERR_TOO_LARGE, "too many values");
    if (!on_device) {   // the host side of this library (make_dev_camera evaluates rtx_tanf there)
        for (size_t i = 0; i < n; ++i) {
            switch (fn) {
                case 0: out[i] = rtx_powf(x[i], y[i]); break;
                case 1: out[i] = rtx_expf(x[i]); break;
                case 2: out[i] = rtx_sinf(x[i]); break;
                case 3: out[i] = rtx_cosf(x[i]); break;
                case 4: out[i] = rtx_tanf(x[i]); break;
                case 5: out[i] = rtx_asinf(x[i]); break;
                case 6: out[i] = rtx_acosf(x[i]); break;
                case 8: out[i] = rtx_unorm8(x[i]); break;
                default: out[i] = rtx_atan2f(x[i], y[i]); break;
            }
        }
        return RTX_OK;
    }
    if (rtx_device_count() <= 0) return rtx_fail(RTX_ERR_NO_DEVICE, "no HIP device present");
    float *dx = nullptr, *dy = nullptr, *dout = nullptr;
    if (!n) return RTX_OK;
    hipError_t e = hipMalloc((void**)&dx, n * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&dy, n * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&dout, n * 4);
    if (e == hipSuccess) e = hipMemcpy(dx, x, n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dy, y, n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        launch_math_probe(fn, dx, dy, (uint32_t)n, dout, nullptr);
        e = hipMemcpy(out, dout, n * 4, hipMemcpyDeviceToHost);
    }
    if (dx) (void)hipFree(dx);
    if (dy) (void)hipFree(dy);
    if (dout) (void)hipFree(dout);
    if (e != hipSuccess) return rtx_fail(RTX_ERR_HIP, std::string("math probe: ") + hipGetErrorString(e));
    return RTX_OK;
}

int rtx_scene_info(const rtx_scene* s, uint64_t counts[6], float bounds[6]) {
    if (!s) return rtx_fail(RTX_ERR_INVALID, "scene is NULL");
    if (counts) {
        counts[0] = s->nTris;
        counts[1] = s->nNodes;
        counts[2] = sizeof(BvhNode);
        counts[3] = sizeof(TriRec);
        counts[4] = s->deviceBytes;
        counts[5] = (uint64_t)s->depth;
    }
    if (bounds) std::memcpy(bounds, s->dev.bounds, sizeof(float) * 6);
    return RTX_OK;
}

int rtx_set_counters(rtx_scene* s, int enabled) {
    if (!s) return rtx_fail(RTX_ERR_INVALID, "scene is NULL");
    s->cfg.counting = enabled ? 1 : 0;
    RTX_HIP(hipMemset(s->cfg.counters, 0, 8 * sizeof(unsigned long long)));
    return RTX_OK;
}

#ifdef TRAV_PROFILE
int rtx_debug_trav_profile(rtx_scene* s, uint64_t out[8]) {   // variant builds only (tools/trav_profile.py)
    RTX_HIP(hipDeviceSynchronize());
    RTX_HIP(hipMemcpy(out, s->cfg.counters, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    RTX_HIP(hipMemset(s->cfg.counters, 0, 8 * sizeof(uint64_t)));
    return RTX_OK;
}
#endif

int rtx_get_counters(rtx_scene* s, uint64_t out[3]) {
    if (!s) return rtx_fail(RTX_ERR_INVALID, "scene is NULL");
    RTX_HIP(hipDeviceSynchronize());
    RTX_HIP(hipMemcpy(out, s->cfg.counters, 3 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return RTX_OK;
}

int rtx_trace_closest_device(rtx_scene* s, const void* rays_dev, size_t n, int hit_side, int with_lights, void* hit4_dev,
                             void* stream) {
    if (!s) return rtx_fail(RTX_ERR_INVALID, "scene is NULL");
    if (hit_side < RTX_HIT_FRONT || hit_side > RTX_HIT_FRONT_AND_BACK) return rtx_fail(RTX_ERR_INVALID, "bad hit_side");
    if (n > 0xFFFFFF00ull) return rtx_fail(RTX_ERR_TOO_LARGE, "more than 2^32 rays in one call");
    if (n && (!rays_dev || !hit4_dev)) return rtx_fail(RTX_ERR_INVALID, "NULL ray/hit buffer");
    LaunchCfg cfg = s->cfg;
    cfg.stream = (hipStream_t)stream;
    launch_extend(rtx_traversal_scene(s), cfg, (const float4*)rays_dev, (uint32_t)n, hit_side, with_lights, (float4*)hit4_dev);
    RTX_HIP(hipGetLastError());
    return RTX_OK;
}

int rtx_trace_closest(rtx_scene* s, const rtx_ray* rays, size_t n, int hit_side, int with_lights, rtx_hit* hits) {
    if (!s) return rtx_fail(RTX_ERR_INVALID, "scene is NULL");
    if (n == 0) return RTX_OK;
    if (!rays || !hits) return rtx_fail(RTX_ERR_INVALID, "NULL ray/hit array");
    int rc;
    if ((rc = s->rays.ensure(n * sizeof(rtx_ray))) != RTX_OK) return rc;
    if ((rc = s->hits.ensure(n * sizeof(float4))) != RTX_OK) return rc;
    if ((rc = s->full.ensure(n * sizeof(rtx_hit))) != RTX_OK) return rc;
    RTX_HIP(hipMemcpy(s->rays.p, rays, n * sizeof(rtx_ray), hipMemcpyHostToDevice));
    if ((rc = rtx_trace_closest_device(s, s->rays.p, n, hit_side, with_lights, s->hits.p, nullptr)) != RTX_OK) return rc;
    launch_expand_hits(s->dev, (const float4*)s->rays.p, (const float4*)s->hits.p, (uint32_t)n, (rtx_hit*)s->full.p, nullptr);
    RTX_HIP(hipGetLastError());
    RTX_HIP(hipMemcpy(hits, s->full.p, n * sizeof(rtx_hit), hipMemcpyDeviceToHost));
    return RTX_OK;
}

int rtx_trace_visibility(rtx_scene* s, const rtx_ray* rays, const float* normals, size_t n, int use_transmittance, float* vis) {
    if (!s) return rtx_fail(RTX_ERR_INVALID, "scene is NULL");
    if (n == 0) return RTX_OK;
    if (!rays || !normals || !vis) return rtx_fail(RTX_ERR_INVALID, "NULL array");
    if (n > 0xFFFFFF00ull) return rtx_fail(RTX_ERR_TOO_LARGE, "more than 2^32 rays in one call");
    int rc;
    if ((rc = s->rays.ensure(n * sizeof(rtx_ray))) != RTX_OK) return rc;
    if ((rc = s->aux.ensure(n * sizeof(float4))) != RTX_OK) return rc;
    if ((rc = s->aux2.ensure(n * sizeof(float))) != RTX_OK) return rc;
    std::vector<float> n4(n * 4);
    for (size_t i = 0; i < n; ++i) {
        n4[4 * i] = normals[3 * i];
        n4[4 * i + 1] = normals[3 * i + 1];
        n4[4 * i + 2] = normals[3 * i + 2];
        n4[4 * i + 3] = 0.0f;
    }
    RTX_HIP(hipMemcpy(s->rays.p, rays, n * sizeof(rtx_ray), hipMemcpyHostToDevice));
    RTX_HIP(hipMemcpy(s->aux.p, n4.data(), n * sizeof(float4), hipMemcpyHostToDevice));
    launch_visibility(rtx_traversal_scene(s), s->cfg, (const float4*)s->rays.p, (const float4*)s->aux.p, (uint32_t)n, use_transmittance,
                      (float*)s->aux2.p, nullptr);
    RTX_HIP(hipGetLastError());
    RTX_HIP(hipMemcpy(vis, s->aux2.p, n * sizeof(float), hipMemcpyDeviceToHost));
    return RTX_OK;
}

int rtx_render_primary(rtx_scene* s, const rtx_camera* cam, int hit_side, int* prim, float* t, unsigned char* front) {
    if (!s) return rtx_fail(RTX_ERR_INVALID, "scene is NULL");
    DevCamera dc;
    int rc;
    if ((rc = make_dev_camera(cam, dc)) != RTX_OK) return rc;
    const size_t n = (size_t)dc.W * dc.H;
    if ((rc = s->rays.ensure(n * sizeof(rtx_ray))) != RTX_OK) return rc;
    if ((rc = s->hits.ensure(n * sizeof(float4))) != RTX_OK) return rc;
    if ((rc = s->full.ensure(n * sizeof(rtx_hit))) != RTX_OK) return rc;
    launch_primary_raygen(dc, (float4*)s->rays.p, nullptr);
    if ((rc = rtx_trace_closest_device(s, s->rays.p, n, hit_side, 0, s->hits.p, nullptr)) != RTX_OK) return rc;
    launch_expand_hits(s->dev, (const float4*)s->rays.p, (const float4*)s->hits.p, (uint32_t)n, (rtx_hit*)s->full.p, nullptr);
    RTX_HIP(hipGetLastError());
    std::vector<rtx_hit> h(n);
    RTX_HIP(hipMemcpy(h.data(), s->full.p, n * sizeof(rtx_hit), hipMemcpyDeviceToHost));
    std::vector<int> base(s->dev.n_objs);
    std::vector<ObjRec> objs(s->dev.n_objs);
    if (s->dev.n_objs) RTX_HIP(hipMemcpy(objs.data(), s->dev.objs, objs.size() * sizeof(ObjRec), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i) {
        const bool hit = h[i].obj >= 0;
        if (prim) prim[i] = hit ? objs[h[i].obj].face_base + h[i].face : -1;
        if (t) t[i] = hit ? h[i].t : RTX_BIGFLOAT;
        if (front) front[i] = hit ? (unsigned char)h[i].front : 1;
    }
    return RTX_OK;
}

}  // extern "C"
