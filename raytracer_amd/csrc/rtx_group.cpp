// C ABI (include/rtx.h): several GPUs driven from one host process. Replaces the worker pool and mergeToImage of
// RayCaster::CastRays (reference RayTracer/RayCaster.cpp:1281-1312, 1182-1212) and the worker threads of
// PhotonMapBuilder::EmitPhotons that feed one map (RayTracer/PhotonMapBuilder.cpp:102-146) across the GPUs of a node.
#include "rtx_internal.h"
#include "photon.h"
#include <algorithm>
#include <chrono>
#include <cstring>
#include <thread>

namespace {

struct Member {
    int device = 0;
    rtx_scene* scene = nullptr;
    rtx_photons* photons = nullptr;
    rtx_film film{};           // on `device`
    rtx_film staged{};         // copy of this member's film on member 0's device (members > 0 on another device)
};

struct FilmAlloc {
    static size_t bytes(int k, size_t n) { return (k == 1 || k == 2) ? n * 4 : n * 16; }
    static void** slot(rtx_film& f, int k) {
        switch (k) {
            case 0: return (void**)&f.accum;
            case 1: return (void**)&f.count;
            case 2: return (void**)&f.z;
            case 3: return (void**)&f.albedo;
            default: return (void**)&f.normal;
        }
    }
    static void release(rtx_film& f) {
        for (int k = 0; k < 5; ++k) {
            void** p = slot(f, k);
            if (*p) (void)hipFree(*p);
            *p = nullptr;
        }
    }
    static int create(rtx_film& f, size_t n) {   // on the current device
        for (int k = 0; k < 5; ++k)
            if (hipMalloc(slot(f, k), bytes(k, n)) != hipSuccess) {
                release(f);
                return rtx_fail(RTX_ERR_HIP, "group film allocation failed");
            }
        return RTX_OK;
    }
};

}  // namespace

struct rtx_group {
    std::vector<Member> m;
    size_t filmPixels = 0;
};

// runs fn(member index) on one host thread per member, each with its member's device current; the first failure's
// message becomes this thread's rtx_last_error
template <class F>
static int for_each_member(rtx_group* g, F fn) {
    const int n = (int)g->m.size();
    std::vector<int> rc(n, RTX_OK);
    std::vector<std::string> err(n);
    auto body = [&](int i) {
        if (hipSetDevice(g->m[i].device) != hipSuccess) {
            rc[i] = RTX_ERR_HIP;
            err[i] = "hipSetDevice failed for group member " + std::to_string(i);
            return;
        }
        rc[i] = fn(i);
        if (rc[i] != RTX_OK) err[i] = rtx_last_error();
    };
    std::vector<std::thread> th;
    for (int i = 1; i < n; ++i) th.emplace_back(body, i);
    int prev = 0;
    (void)hipGetDevice(&prev);
    body(0);
    for (auto& t : th) t.join();
    (void)hipSetDevice(prev);
    for (int i = 0; i < n; ++i)
        if (rc[i] != RTX_OK) return rtx_fail(rc[i], "group member " + std::to_string(i) + " (device " + std::to_string(g->m[i].device) + "): " + err[i]);
    return RTX_OK;
}

static void release_films(rtx_group* g) {
    int prev = 0;
    (void)hipGetDevice(&prev);
    for (size_t i = 0; i < g->m.size(); ++i) {
        (void)hipSetDevice(g->m[i].device);
        FilmAlloc::release(g->m[i].film);
        (void)hipSetDevice(g->m[0].device);
        FilmAlloc::release(g->m[i].staged);
    }
    (void)hipSetDevice(prev);
    g->filmPixels = 0;
}

static int ensure_films(rtx_group* g, size_t n) {
    if (g->filmPixels == n) return RTX_OK;
    release_films(g);
    int prev = 0;
    (void)hipGetDevice(&prev);
    int rc = RTX_OK;
    for (size_t i = 0; i < g->m.size() && rc == RTX_OK; ++i) {
        RTX_HIP(hipSetDevice(g->m[i].device));
        rc = FilmAlloc::create(g->m[i].film, n);
        if (rc == RTX_OK && i > 0 && g->m[i].device != g->m[0].device) {
            RTX_HIP(hipSetDevice(g->m[0].device));
            rc = FilmAlloc::create(g->m[i].staged, n);
        }
    }
    (void)hipSetDevice(prev);
    if (rc != RTX_OK) {
        release_films(g);
        return rc;
    }
    g->filmPixels = n;
    return RTX_OK;
}

extern "C" {

void rtx_group_destroy(rtx_group* g) {
    if (!g) return;
    int prev = 0;
    (void)hipGetDevice(&prev);
    if (!g->m.empty()) release_films(g);
    for (Member& mb : g->m) {
        (void)hipSetDevice(mb.device);
        rtx_photons_destroy(mb.photons);
        rtx_scene_destroy(mb.scene);
    }
    (void)hipSetDevice(prev);
    delete g;
}

int rtx_group_create(const rtx_scene_desc* desc, const int* device_ids, int n, rtx_group** out) {
    if (!out) return rtx_fail(RTX_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!desc || !device_ids || n < 1 || n > RTX_GROUP_MAX) return rtx_fail(RTX_ERR_INVALID, "need 1 <= n <= RTX_GROUP_MAX members");
    const int nd = rtx_device_count();
    if (nd <= 0) return rtx_fail(RTX_ERR_NO_DEVICE, "no HIP device present (this library has no CPU path)");
    for (int i = 0; i < n; ++i)
        if (device_ids[i] < 0 || device_ids[i] >= nd) return rtx_fail(RTX_ERR_INVALID, "device id " + std::to_string(device_ids[i]) + " out of range");
    rtx_group* g = new rtx_group();
    g->m.resize(n);
    for (int i = 0; i < n; ++i) g->m[i].device = device_ids[i];
    // one replica per member, built concurrently (the BVH build is host work)
    int rc = for_each_member(g, [&](int i) { return rtx_scene_create(desc, &g->m[i].scene); });
    if (rc != RTX_OK) {
        const std::string msg = rtx_last_error();
        rtx_group_destroy(g);
        return rtx_fail(rc, msg);
    }
    *out = g;
    return RTX_OK;
}

int rtx_group_size(const rtx_group* g) { return g ? (int)g->m.size() : 0; }
rtx_scene* rtx_group_scene(rtx_group* g, int member) { return (g && member >= 0 && member < (int)g->m.size()) ? g->m[member].scene : nullptr; }
rtx_photons* rtx_group_photons(rtx_group* g, int member) { return (g && member >= 0 && member < (int)g->m.size()) ? g->m[member].photons : nullptr; }

int rtx_group_cancel(rtx_group* g) {
    if (!g) return rtx_fail(RTX_ERR_INVALID, "group is NULL");
    for (Member& mb : g->m) (void)rtx_cancel(mb.scene);
    return RTX_OK;
}

int rtx_group_render(rtx_group* g, const rtx_camera* cam, const rtx_render_params* params, float* linear_rgb, unsigned char* rgb8,
                     float* z, int* samples, float* albedo, float* normal, rtx_stats* stats) {
    if (!g || !cam || !params) return rtx_fail(RTX_ERR_INVALID, "NULL argument");
    if (cam->width <= 0 || cam->height <= 0) return rtx_fail(RTX_ERR_INVALID, "invalid camera");
    const int n = (int)g->m.size();
    const size_t np = (size_t)cam->width * cam->height;
    int rc = ensure_films(g, np);
    if (rc != RTX_OK) return rc;
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<rtx_stats> st(n);
    rc = for_each_member(g, [&](int i) {
        Member& mb = g->m[i];
        rtx_render_params rp = *params;
        rp.rank = i;
        rp.world = n;
        int r = rtx_film_clear(&mb.film, cam->width, cam->height, nullptr);
        if (r == RTX_OK) r = rtx_render_accumulate(mb.scene, cam, &rp, (params->use_photons ? mb.photons : nullptr), &mb.film, &st[i]);
        if (r != RTX_OK) return r;
        // this member's film onto member 0's device: every peer sends over its own xGMI link at the same time
        if (i > 0 && mb.device != g->m[0].device)
            for (int k = 0; k < 5; ++k)
                RTX_HIP(hipMemcpyPeer(*FilmAlloc::slot(mb.staged, k), g->m[0].device, *FilmAlloc::slot(mb.film, k), mb.device, FilmAlloc::bytes(k, np)));
        RTX_HIP(hipDeviceSynchronize());
        return (int)RTX_OK;
    });
    if (rc != RTX_OK) return rc;
    int prev = 0;
    (void)hipGetDevice(&prev);
    RTX_HIP(hipSetDevice(g->m[0].device));
    if (n > 1) {
        rtx_film srcs[RTX_GROUP_MAX - 1];
        for (int i = 1; i < n; ++i) srcs[i - 1] = g->m[i].device != g->m[0].device ? g->m[i].staged : g->m[i].film;
        launch_film_merge(g->m[0].film, srcs, n - 1, (uint32_t)np, nullptr);
        RTX_HIP(hipGetLastError());
    }
    rc = rtx_film_resolve(&g->m[0].film, cam, linear_rgb, rgb8, z, samples, albedo, normal);
    (void)hipSetDevice(prev);
    if (rc != RTX_OK) return rc;
    if (stats) {
        rtx_stats a;
        std::memset(&a, 0, sizeof(a));
        for (const rtx_stats& s : st) {
            a.rays_primary += s.rays_primary;
            a.rays_bounce += s.rays_bounce;
            a.rays_shadow += s.rays_shadow;
            a.paths += s.paths;
            a.nodes_visited += s.nodes_visited;
            a.tris_tested += s.tris_tested;
            a.extend_launches += s.extend_launches;
            a.rays_extend += s.rays_extend;
            a.extend_ms = std::max(a.extend_ms, s.extend_ms);
            a.shadow_ms = std::max(a.shadow_ms, s.shadow_ms);
            a.shade_ms = std::max(a.shade_ms, s.shade_ms);
        }
        a.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();   // wall, merge and resolve included
        *stats = a;
    }
    return RTX_OK;
}

int rtx_group_photon_build(rtx_group* g, const rtx_photon_params* params, rtx_stats* stats) {
    if (!g || !params) return rtx_fail(RTX_ERR_INVALID, "NULL argument");
    const int n = (int)g->m.size();
    std::vector<rtx_stats> st(n);
    std::vector<rtx_photons*> fresh(n, nullptr);
    int rc = for_each_member(g, [&](int i) {
        rtx_photon_params p = *params;
        p.first_photon = i;
        p.photon_stride = n;
        return rtx_photon_build(g->m[i].scene, &p, &fresh[i], &st[i]);
    });
    if (rc == RTX_OK && n > 1) {
        // all-gather: every member's GPU receives the raw records of all members, in member order (the same map everywhere),
        // before any set is replaced (rtx_photon_set may reallocate the buffer the others still read)
        std::vector<void*> gathered(2 * (size_t)n, nullptr);
        size_t total[2] = {0, 0};
        for (int w = 0; w < 2; ++w)
            for (int i = 0; i < n; ++i) total[w] += fresh[i]->set[w].n;
        rc = for_each_member(g, [&](int d) {
            for (int w = 0; w < 2; ++w) {
                if (!total[w]) continue;
                RTX_HIP(hipMalloc(&gathered[2 * (size_t)d + w], total[w] * sizeof(DevPhotonRec)));
                size_t off = 0;
                for (int i = 0; i < n; ++i) {
                    const size_t cnt = fresh[i]->set[w].n;
                    if (cnt)
                        RTX_HIP(hipMemcpyPeer((char*)gathered[2 * (size_t)d + w] + off * sizeof(DevPhotonRec), g->m[d].device, fresh[i]->set[w].raw,
                                              g->m[i].device, cnt * sizeof(DevPhotonRec)));
                    off += cnt;
                }
            }
            RTX_HIP(hipDeviceSynchronize());
            return (int)RTX_OK;
        });
        if (rc == RTX_OK)
            rc = for_each_member(g, [&](int d) {
                for (int w = 0; w < 2; ++w) {
                    const int r = rtx_photon_set(fresh[d], w, gathered[2 * (size_t)d + w], total[w], params->num_photons);
                    if (r != RTX_OK) return r;
                }
                return (int)RTX_OK;
            });
        const std::string msg = rc != RTX_OK ? rtx_last_error() : "";
        (void)for_each_member(g, [&](int d) {
            for (int w = 0; w < 2; ++w)
                if (gathered[2 * (size_t)d + w]) (void)hipFree(gathered[2 * (size_t)d + w]);
            return (int)RTX_OK;
        });
        if (rc != RTX_OK) rtx_set_error(msg);
    }
    if (rc != RTX_OK) {
        const std::string msg = rtx_last_error();
        (void)for_each_member(g, [&](int i) {
            rtx_photons_destroy(fresh[i]);
            return (int)RTX_OK;
        });
        return rtx_fail(rc, msg);
    }
    (void)for_each_member(g, [&](int i) {
        rtx_photons_destroy(g->m[i].photons);
        g->m[i].photons = fresh[i];
        return (int)RTX_OK;
    });
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        for (const rtx_stats& s : st) {
            stats->rays_photon += s.rays_photon;
            stats->total_ms = std::max(stats->total_ms, s.total_ms);
        }
    }
    return RTX_OK;
}

}  // extern "C"
