// C ABI (include/rtx.h): several GPUs driven from one host process. Replaces the worker pool and mergeToImage of
// RayCaster::CastRays (reference RayTracer/RayCaster.cpp:1281-1312, 1182-1212) and the worker threads of
// PhotonMapBuilder::EmitPhotons that feed one map (RayTracer/PhotonMapBuilder.cpp:102-146) across the GPUs of a node.
#include "rtx_internal.h"
#include "photon.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace {

struct Member {
    int device = 0;
    rtx_scene* scene = nullptr;
    rtx_photons* photons = nullptr;
    rtx_film film{};           // on `device`
    rtx_film staged{};         // copy of this member's film on member 0's device (members > 0 on another device)
    // adaptive rounds (spp_max > spp_min, more than one member): per-pixel statistics on `device`, and for members on
    // another device their copy on member 0's
    rtx_pixel_stats pstats{};
    rtx_pixel_stats pstatsStaged{};
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
    size_t statPixels = 0;
    int* listDev = nullptr;    // on member 0's device: the pixel list of one member (statPixels entries)
    // RayCaster::Stop() for the whole group: set by rtx_group_cancel, cleared when rtx_group_render starts. The members' own
    // flags (rtx_cancel) stop the pass in flight; they are cleared again by every rtx_render_round, so the loop over adaptive
    // rounds needs a flag of its own that outlives a round.
    std::atomic<int> cancel{0};
};

static void free_stats(rtx_pixel_stats& p) {
    if (p.sum) (void)hipFree(p.sum);
    if (p.sum2) (void)hipFree(p.sum2);
    p.sum = p.sum2 = nullptr;
}
static int alloc_stats(rtx_pixel_stats& p, size_t n) {   // on the current device
    if (hipMalloc((void**)&p.sum, n * 16) != hipSuccess || hipMalloc((void**)&p.sum2, n * 16) != hipSuccess) {
        free_stats(p);
        return rtx_fail(RTX_ERR_HIP, "group statistics allocation failed");
    }
    return RTX_OK;
}
static void release_stats(rtx_group* g) {
    int prev = 0;
    (void)hipGetDevice(&prev);
    for (size_t i = 0; i < g->m.size(); ++i) {
        (void)hipSetDevice(g->m[i].device);
        free_stats(g->m[i].pstats);
        (void)hipSetDevice(g->m[0].device);
        free_stats(g->m[i].pstatsStaged);
    }
    if (!g->m.empty()) (void)hipSetDevice(g->m[0].device);
    if (g->listDev) (void)hipFree(g->listDev);
    g->listDev = nullptr;
    (void)hipSetDevice(prev);
    g->statPixels = 0;
}
static int ensure_stats(rtx_group* g, size_t n) {
    if (g->statPixels == n) return RTX_OK;
    release_stats(g);
    int prev = 0;
    (void)hipGetDevice(&prev);
    int rc = RTX_OK;
    for (size_t i = 0; i < g->m.size() && rc == RTX_OK; ++i) {
        if (hipSetDevice(g->m[i].device) != hipSuccess) rc = rtx_fail(RTX_ERR_HIP, "hipSetDevice failed");
        if (rc == RTX_OK) rc = alloc_stats(g->m[i].pstats, n);
        if (rc == RTX_OK && i > 0 && g->m[i].device != g->m[0].device) {
            if (hipSetDevice(g->m[0].device) != hipSuccess) rc = rtx_fail(RTX_ERR_HIP, "hipSetDevice failed");
            if (rc == RTX_OK) rc = alloc_stats(g->m[i].pstatsStaged, n);
        }
    }
    if (rc == RTX_OK && (hipSetDevice(g->m[0].device) != hipSuccess || hipMalloc((void**)&g->listDev, n * sizeof(int)) != hipSuccess))
        rc = rtx_fail(RTX_ERR_HIP, "group pixel list allocation failed");
    (void)hipSetDevice(prev);
    if (rc != RTX_OK) {
        const std::string msg = rtx_last_error();
        release_stats(g);
        return rtx_fail(rc, msg);
    }
    g->statPixels = n;
    return RTX_OK;
}

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
        if (hipSetDevice(g->m[i].device) != hipSuccess) rc = rtx_fail(RTX_ERR_HIP, "hipSetDevice failed for group member " + std::to_string(i));
        if (rc == RTX_OK) rc = FilmAlloc::create(g->m[i].film, n);
        if (rc == RTX_OK && i > 0 && g->m[i].device != g->m[0].device) {
            if (hipSetDevice(g->m[0].device) != hipSuccess) rc = rtx_fail(RTX_ERR_HIP, "hipSetDevice failed for group member 0");
            if (rc == RTX_OK) rc = FilmAlloc::create(g->m[i].staged, n);
        }
    }
    (void)hipSetDevice(prev);   // the caller's device comes back and the films created so far are released on every way out
    if (rc != RTX_OK) {
        const std::string msg = rtx_last_error();
        release_films(g);
        return rtx_fail(rc, msg);
    }
    g->filmPixels = n;
    return RTX_OK;
}

static void add_stats(rtx_stats& a, const rtx_stats& s) {
    a.rays_primary += s.rays_primary;
    a.rays_bounce += s.rays_bounce;
    a.rays_shadow += s.rays_shadow;
    a.paths += s.paths;
    a.nodes_visited += s.nodes_visited;
    a.tris_tested += s.tris_tested;
    a.extend_launches += s.extend_launches;
    a.rays_extend += s.rays_extend;
    a.shade_launches += s.shade_launches;
    a.vertices += s.vertices;
    a.shadow_launches += s.shadow_launches;
    a.shadow_segments += s.shadow_segments;
    a.shadow_nodes += s.shadow_nodes;
    a.shadow_tris += s.shadow_tris;
    a.nee_ms += s.nee_ms;
    a.repacks += s.repacks;
    a.extend_ms += s.extend_ms;
    a.shadow_ms += s.shadow_ms;
    a.shade_ms += s.shade_ms;
    a.total_ms += s.total_ms;
}

// Greedy longest-processing-time deal of the buckets that still hold unconverged pixels: heaviest bucket first (ties by
// bucket index) to the least loaded member (ties to the lowest member) - the same rule on every run, and the rule of
// raytracer_amd/distributed.py::lpt_assign.
static void lpt_assign(const std::vector<int>& weight, int members, std::vector<int>& owner) {
    owner.assign(weight.size(), -1);
    std::vector<int> busy;
    for (size_t b = 0; b < weight.size(); ++b)
        if (weight[b] > 0) busy.push_back((int)b);
    std::stable_sort(busy.begin(), busy.end(), [&](int x, int y) { return weight[x] > weight[y]; });
    std::vector<long long> load(members, 0);
    for (int b : busy) {
        int r = 0;
        for (int k = 1; k < members; ++k)
            if (load[k] < load[r]) r = k;
        owner[b] = r;
        load[r] += weight[b];
    }
}

// Adaptive CastRays (spp_min ... spp_max, RayCaster.cpp:770-980) over the members with the work dealt out again after
// every round: round 1 on the static split b % n; after a round the statistics of the pixels each member
// sampled are collected on member 0's GPU, the confidence-interval test runs there, the buckets that still hold
// unconverged pixels are dealt out by lpt_assign, and the merged statistics go back to every member (a bucket that
// changes hands continues from the sums its previous owner left). Accumulated colour never moves: it stays in the film
// of the member that traced it and the films are summed at frame end, as in the fixed-spp case.
static int group_render_adaptive(rtx_group* g, const rtx_camera* cam, const rtx_render_params* params, std::vector<rtx_stats>& st) {
    const int n = (int)g->m.size();
    const int W = cam->width, H = cam->height;
    const size_t np = (size_t)W * H;
    const int tile = params->tile_size > 0 ? params->tile_size : 16;
    int rc = ensure_stats(g, np);
    if (rc != RTX_OK) return rc;
    std::vector<std::vector<int>> mine(n);
    for (int i = 0; i < n; ++i) {
        const int64_t len = rtx_debug_pixel_list(W, H, tile, i, n, nullptr, 0);
        if (len < 0) return (int)len;
        std::vector<int> padded((size_t)len);
        (void)rtx_debug_pixel_list(W, H, tile, i, n, padded.data(), padded.size());
        for (int p : padded)
            if (p >= 0) mine[i].push_back(p);
    }
    rc = for_each_member(g, [&](int i) {
        Member& mb = g->m[i];
        int r = rtx_film_clear(&mb.film, W, H, nullptr);
        if (r != RTX_OK) return r;
        RTX_HIP(hipMemset(mb.pstats.sum, 0, np * 16));
        RTX_HIP(hipMemset(mb.pstats.sum2, 0, np * 16));
        RTX_HIP(hipDeviceSynchronize());
        return (int)RTX_OK;
    });
    if (rc != RTX_OK) return rc;
    const int nbx = (W + tile - 1) / tile, nby = (H + tile - 1) / tile;
    std::vector<int> refine(np), weight, owner;
    int base = 0;
    for (int round = 0;; ++round) {
        // Stop(): no further round is dealt; the films hold the rounds that completed (a round cancelled in flight drops its
        // unfinished pass, as rtx_render_accumulate does) and are merged as they are
        if (g->cancel.load(std::memory_order_relaxed)) break;
        const int nS = std::min(params->spp_min, params->spp_max - base);
        if (nS <= 0) break;
        const bool more = params->spp_max > params->spp_min && base + nS < params->spp_max;
        rc = for_each_member(g, [&](int i) {
            Member& mb = g->m[i];
            rtx_stats one;
            std::memset(&one, 0, sizeof(one));
            if (!mine[i].empty() && !g->cancel.load(std::memory_order_relaxed)) {   // (a Stop() that came before this member's round began)
                const int r = rtx_render_round(mb.scene, cam, params, (params->use_photons ? mb.photons : nullptr), &mb.film, &mb.pstats,
                                               mine[i].data(), mine[i].size(), base, nS, round == 1 ? 1 : 0, &one);
                if (r != RTX_OK) return r;
                add_stats(st[i], one);
            }
            // statistics towards member 0's device, every peer over its own link
            if (more && i > 0 && mb.device != g->m[0].device) {
                RTX_HIP(hipMemcpyPeer(mb.pstatsStaged.sum, g->m[0].device, mb.pstats.sum, mb.device, np * 16));
                RTX_HIP(hipMemcpyPeer(mb.pstatsStaged.sum2, g->m[0].device, mb.pstats.sum2, mb.device, np * 16));
            }
            RTX_HIP(hipDeviceSynchronize());
            return (int)RTX_OK;
        });
        if (rc != RTX_OK) return rc;
        if (g->cancel.load(std::memory_order_relaxed)) break;   // the round just dealt was interrupted: its statistics are not to be trusted, nothing is dealt from them
        base += nS;
        if (!more) break;
        // ---- on member 0's device: collect, test, deal ----
        struct DeviceGuard {   // the caller's device comes back on every way out of this block
            int prev = 0;
            DeviceGuard() { (void)hipGetDevice(&prev); }
            ~DeviceGuard() { (void)hipSetDevice(prev); }
        } guard;
        RTX_HIP(hipSetDevice(g->m[0].device));
        for (int i = 1; i < n; ++i) {
            if (mine[i].empty()) continue;
            const rtx_pixel_stats& src = g->m[i].device != g->m[0].device ? g->m[i].pstatsStaged : g->m[i].pstats;
            RTX_HIP(hipMemcpy(g->listDev, mine[i].data(), mine[i].size() * sizeof(int), hipMemcpyHostToDevice));
            launch_stats_take(g->m[0].pstats.sum, g->m[0].pstats.sum2, src.sum, src.sum2, g->listDev, (uint32_t)mine[i].size(), nullptr);
            RTX_HIP(hipGetLastError());
            RTX_HIP(hipDeviceSynchronize());   // listDev is reused by the next member
        }
        const int64_t nRef = rtx_refine_list(&g->m[0].pstats, W, H, params->ci_halfwidth, params->spp_max, refine.data(), refine.size());
        (void)hipSetDevice(guard.prev);
        if (nRef < 0) return (int)nRef;
        if (nRef == 0) break;
        weight.assign((size_t)nbx * nby, 0);
        for (int64_t k = 0; k < nRef; ++k) weight[(size_t)((refine[k] / W) / tile) * nbx + (refine[k] % W) / tile]++;
        lpt_assign(weight, n, owner);
        for (int i = 0; i < n; ++i) mine[i].clear();
        for (int64_t k = 0; k < nRef; ++k) mine[owner[(size_t)((refine[k] / W) / tile) * nbx + (refine[k] % W) / tile]].push_back(refine[k]);
        // ---- the merged statistics back to the members ----
        rc = for_each_member(g, [&](int i) {
            Member& mb = g->m[i];
            if (i == 0 || mine[i].empty()) return (int)RTX_OK;
            if (mb.device != g->m[0].device) {
                RTX_HIP(hipMemcpyPeer(mb.pstats.sum, mb.device, g->m[0].pstats.sum, g->m[0].device, np * 16));
                RTX_HIP(hipMemcpyPeer(mb.pstats.sum2, mb.device, g->m[0].pstats.sum2, g->m[0].device, np * 16));
            } else {
                RTX_HIP(hipMemcpy(mb.pstats.sum, g->m[0].pstats.sum, np * 16, hipMemcpyDeviceToDevice));
                RTX_HIP(hipMemcpy(mb.pstats.sum2, g->m[0].pstats.sum2, np * 16, hipMemcpyDeviceToDevice));
            }
            RTX_HIP(hipDeviceSynchronize());
            return (int)RTX_OK;
        });
        if (rc != RTX_OK) return rc;
    }
    // films towards member 0's device
    return for_each_member(g, [&](int i) {
        Member& mb = g->m[i];
        if (i > 0 && mb.device != g->m[0].device)
            for (int k = 0; k < 5; ++k)
                RTX_HIP(hipMemcpyPeer(*FilmAlloc::slot(mb.staged, k), g->m[0].device, *FilmAlloc::slot(mb.film, k), mb.device, FilmAlloc::bytes(k, np)));
        RTX_HIP(hipDeviceSynchronize());
        return (int)RTX_OK;
    });
}

extern "C" {

void rtx_group_destroy(rtx_group* g) {
    if (!g) return;
    int prev = 0;
    (void)hipGetDevice(&prev);
    if (!g->m.empty()) {
        release_films(g);
        release_stats(g);
    }
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

int rtx_debug_lpt_assign(const int* weight, size_t n_buckets, int members, int* owner) {
    if (!weight || !owner || members < 1 || members > RTX_GROUP_MAX) return rtx_fail(RTX_ERR_INVALID, "bad argument");
    std::vector<int> w(weight, weight + n_buckets), o;
    lpt_assign(w, members, o);
    std::copy(o.begin(), o.end(), owner);
    return RTX_OK;
}

int rtx_group_size(const rtx_group* g) { return g ? (int)g->m.size() : 0; }
rtx_scene* rtx_group_scene(rtx_group* g, int member) { return (g && member >= 0 && member < (int)g->m.size()) ? g->m[member].scene : nullptr; }
rtx_photons* rtx_group_photons(rtx_group* g, int member) { return (g && member >= 0 && member < (int)g->m.size()) ? g->m[member].photons : nullptr; }

int rtx_group_cancel(rtx_group* g) {
    if (!g) return rtx_fail(RTX_ERR_INVALID, "group is NULL");
    g->cancel.store(1);
    for (Member& mb : g->m) (void)rtx_cancel(mb.scene);
    return RTX_OK;
}

int rtx_group_render(rtx_group* g, const rtx_camera* cam, const rtx_render_params* params, float* linear_rgb, unsigned char* rgb8,
                     float* z, int* samples, float* albedo, float* normal, rtx_stats* stats) {
    if (!g || !cam || !params) return rtx_fail(RTX_ERR_INVALID, "NULL argument");
    if (cam->width <= 0 || cam->height <= 0) return rtx_fail(RTX_ERR_INVALID, "invalid camera");
    const int n = (int)g->m.size();
    const size_t np = (size_t)cam->width * cam->height;
    g->cancel.store(0);
    int rc = ensure_films(g, np);
    if (rc != RTX_OK) return rc;
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<rtx_stats> st(n);
    for (rtx_stats& x : st) std::memset(&x, 0, sizeof(x));
    const bool rebalance = n > 1 && params->spp_max > params->spp_min && !std::getenv("RTX_AMD_GROUP_STATIC");
    rc = rebalance ? group_render_adaptive(g, cam, params, st) : for_each_member(g, [&](int i) {
        Member& mb = g->m[i];
        rtx_render_params rp = *params;
        rp.rank = i;
        rp.world = n;
        int r = rtx_film_clear(&mb.film, cam->width, cam->height, nullptr);
        if (r == RTX_OK && !g->cancel.load(std::memory_order_relaxed)) r = rtx_render_accumulate(mb.scene, cam, &rp, (params->use_photons ? mb.photons : nullptr), &mb.film, &st[i]);
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
            a.shade_launches += s.shade_launches;
            a.vertices += s.vertices;
            a.shadow_launches += s.shadow_launches;
            a.shadow_segments += s.shadow_segments;
            a.shadow_nodes += s.shadow_nodes;
            a.shadow_tris += s.shadow_tris;
            a.nee_ms = std::max(a.nee_ms, s.nee_ms);
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
