// C ABI (include/rtx.h): photon mapping — PhotonMapBuilder::BuildPhotonMap on the GPU
// (reference RayTracer/PhotonMapBuilder.cpp:20-78).
#include "rtx_internal.h"
#include "photon.h"
#include <algorithm>
#include <cmath>
#include <cstring>

void launch_photon_emit(const DevScene& sc, const LaunchCfg& cfg, int numPhotons, int first, int stride, int maxBounces,
                        unsigned long long seed, DevPhotonRec* g, uint32_t gcap, DevPhotonRec* c, uint32_t ccap, uint32_t* counts);
void launch_photon_bbox(const PhotonSet& s, uint32_t* out6, hipStream_t st);
void launch_photon_count(const PhotonSet& s, uint32_t* cellCount, hipStream_t st);
int launch_photon_order(const PhotonSet& s, float scale, hipStream_t st);   // kernels_sort.hip
int launch_photon_keep_smallest(PhotonSet& s, uint32_t keep, hipStream_t st);
void launch_photon_probe(const DevPhotonMap& pm, const float* pos, const float* nrm, uint32_t n, float* out, hipStream_t st);

void photon_device_map(const rtx_photons* p, int which, DevPhotonMap& out) {
    std::memset(&out, 0, sizeof(out));
    const PhotonSet& s = p->set[which];
    if (!s.n || !s.cellStart) return;
    out.pos = s.pos;
    out.dir = s.dir;
    out.power = s.power;
    out.cellStart = s.cellStart;
    std::memcpy(out.imin, s.imin, sizeof(out.imin));
    out.cell = s.cell;
    out.gx = s.gx;
    out.gy = s.gy;
    out.gz = s.gz;
    out.n = (int)s.n;
    out.radius = s.radius;
    out.maxPhotons = p->searchCount <= 30 ? 30 : (p->searchCount <= 50 ? 50 : 100);   // RayCaster.cpp:149-151
}

// Capacity of the reference's maps (PhotonMapBuilder.cpp:27-28): numPhotons for the global map, numPhotons / 10 for the
// caustics map; a full map takes no more photons (cyPhotonMap is absent: parity unpinned)
static uint32_t map_capacity(int which, int numPhotons) {
    const int n = std::max(0, numPhotons);
    return (uint32_t)(which == 0 ? n : n / 10);
}
static int clamp_to_capacity(rtx_photons* p, int which, int numPhotons) {
    PhotonSet& s = p->set[which];
    if (launch_photon_keep_smallest(s, map_capacity(which, numPhotons), nullptr) != 0) return rtx_fail(RTX_ERR_HIP, "photon map capacity clamp failed");
    return RTX_OK;
}

static void free_sorted(PhotonSet& s) {
    if (s.pos) (void)hipFree(s.pos);
    if (s.dir) (void)hipFree(s.dir);
    if (s.power) (void)hipFree(s.power);
    if (s.cellStart) (void)hipFree(s.cellStart);
    s.pos = s.dir = s.power = nullptr;
    s.cellStart = nullptr;
}

// Scale powers, bin the photons into a uniform grid (cell counts + a radix sort by (cell, photon key)).
static int finalize_set(rtx_photons* p, int which, float scale) {
    PhotonSet& s = p->set[which];
    free_sorted(s);
    if (!s.n) return RTX_OK;
    // The grid is a window of the lattice floor(p / cell) anchored at the origin. Its edge is the search radius doubled
    // until the photons' own bounding box spans at most 512 cells per axis and 2^24 in all: a definition that depends on
    // nothing but the photon records, which the oracle holds bit for bit, so both sides bin - and sum - alike.
    uint32_t* dbox = nullptr;
    uint32_t hbox[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
    RTX_HIP(hipMalloc((void**)&dbox, sizeof(hbox)));
    RTX_HIP(hipMemcpy(dbox, hbox, sizeof(hbox), hipMemcpyHostToDevice));
    launch_photon_bbox(s, dbox, nullptr);
    RTX_HIP(hipMemcpy(hbox, dbox, sizeof(hbox), hipMemcpyDeviceToHost));
    (void)hipFree(dbox);
    float lo[3], hi[3];
    for (int a = 0; a < 6; ++a) {
        const uint32_t u = (hbox[a] & 0x80000000u) ? (hbox[a] & 0x7FFFFFFFu) : ~hbox[a];
        std::memcpy(a < 3 ? &lo[a] : &hi[a - 3], &u, 4);
    }
    float cell = std::max(s.radius, 1e-6f);
    for (;; cell *= 2.0f) {
        double total = 1.0;
        bool ok = true;
        for (int a = 0; a < 3; ++a) {
            const float f0 = std::floor(lo[a] / cell), f1 = std::floor(hi[a] / cell);
            const double d = (double)f1 - (double)f0 + 1.0;
            if (!(d <= 512.0) || !(std::fabs(f0) < 4194304.0f) || !(std::fabs(f1) < 4194304.0f)) ok = false;
            total *= d;
        }
        if (ok && total <= 16777216.0) break;
        if (!(cell < 1e30f)) return rtx_fail(RTX_ERR_INVALID, "photon positions are not finite");
    }
    s.cell = cell;
    int g[3];
    for (int a = 0; a < 3; ++a) {
        s.imin[a] = (int)std::floor(lo[a] / cell);
        g[a] = (int)std::floor(hi[a] / cell) - s.imin[a] + 1;
    }
    s.gx = g[0];
    s.gy = g[1];
    s.gz = g[2];
    s.cells = (size_t)s.gx * s.gy * s.gz;
    RTX_HIP(hipMalloc((void**)&s.pos, (size_t)s.n * sizeof(float4)));
    RTX_HIP(hipMalloc((void**)&s.dir, (size_t)s.n * sizeof(float4)));
    RTX_HIP(hipMalloc((void**)&s.power, (size_t)s.n * sizeof(float4)));
    RTX_HIP(hipMalloc((void**)&s.cellStart, (s.cells + 1) * sizeof(uint32_t)));
    uint32_t* cursor = nullptr;
    RTX_HIP(hipMalloc((void**)&cursor, (s.cells + 1) * sizeof(uint32_t)));
    RTX_HIP(hipMemset(cursor, 0, (s.cells + 1) * sizeof(uint32_t)));
    launch_photon_count(s, cursor, nullptr);
    // exclusive scan on the host (a few million cells at most; once per photon build)
    std::vector<uint32_t> cnt(s.cells + 1);
    RTX_HIP(hipMemcpy(cnt.data(), cursor, (s.cells + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost));
    uint32_t run = 0;
    for (size_t i = 0; i <= s.cells; ++i) {
        const uint32_t c = cnt[i];
        cnt[i] = run;
        run += c;
    }
    RTX_HIP(hipMemcpy(s.cellStart, cnt.data(), (s.cells + 1) * sizeof(uint32_t), hipMemcpyHostToDevice));
    (void)hipFree(cursor);
    // records in (cell, photon key) order: the same map whatever order the emit kernel's lanes stored them in
    if (launch_photon_order(s, scale, nullptr) != 0) return rtx_fail(RTX_ERR_HIP, "ordering the photon map failed");
    RTX_HIP(hipDeviceSynchronize());
    return RTX_OK;
}

extern "C" {

void rtx_photons_destroy(rtx_photons* p) {
    if (!p) return;
    for (int k = 0; k < 2; ++k) {
        free_sorted(p->set[k]);
        if (p->set[k].raw) (void)hipFree(p->set[k].raw);
    }
    if (p->counts) (void)hipFree(p->counts);
    delete p;
}

int rtx_photon_build(rtx_scene* s, const rtx_photon_params* pp, rtx_photons** out, rtx_stats* stats) {
    if (!s || !pp || !out) return rtx_fail(RTX_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (pp->num_photons < 0 || pp->max_bounces < 0 || pp->max_bounces > 3) return rtx_fail(RTX_ERR_INVALID, "bad photon parameters");
    if ((long long)pp->num_photons * 4 > 0xFFFFFFFFLL) return rtx_fail(RTX_ERR_TOO_LARGE, "more than 2^30 photons");
    if (pp->first_photon < 0) return rtx_fail(RTX_ERR_INVALID, "first_photon must be >= 0");
    // <= 0 selects the scene-derived radius; NaN / infinity would reach the grid dimensions
    if (!std::isfinite(pp->photon_radius) || !std::isfinite(pp->caustic_radius))
        return rtx_fail(RTX_ERR_INVALID, "photon_radius / caustic_radius must be finite");
    const int stride = pp->photon_stride > 0 ? pp->photon_stride : 1;
    rtx_photons* p = new rtx_photons();
    p->scene = s;
    p->searchCount = pp->search_count > 0 ? pp->search_count : 100;
    p->numPhotons = pp->num_photons;
    // AdjustPhotonGatherSettings (PhotonMapBuilder.cpp:62-78) on rootNode.GetChildBoundBox()
    float photonRadius = 0.2f, causticRadius = 0.05f;   // PhotonMapBuilder.h:47-48
    const float* b = s->dev.bounds;
    if (b[0] <= b[3] && b[1] <= b[4] && b[2] <= b[5]) {
        const float maxExtent = std::max(std::fabs(b[3] - b[0]), std::max(std::fabs(b[4] - b[1]), std::fabs(b[5] - b[2])));
        if (maxExtent > 0.0f) {
            photonRadius = pp->photon_radius <= 0.0f ? std::max(0.05f, maxExtent * 0.02f) : pp->photon_radius;
            causticRadius = pp->caustic_radius <= 0.0f ? std::max(0.01f, photonRadius * 0.25f) : pp->caustic_radius;
        }
    }
    p->set[0].radius = photonRadius;
    p->set[1].radius = causticRadius;
    const uint32_t nMine = (pp->num_photons > pp->first_photon) ? (uint32_t)((pp->num_photons - pp->first_photon + stride - 1) / stride) : 0u;
    // a photon stores at most once per bounce >= 1
    const uint32_t cap = std::max<uint32_t>(1u, nMine * (uint32_t)std::max(1, pp->max_bounces));
    int rc = RTX_OK;
    for (int k = 0; k < 2 && rc == RTX_OK; ++k) {
        p->set[k].capacity = cap;
        if (hipMalloc((void**)&p->set[k].raw, (size_t)cap * sizeof(DevPhotonRec)) != hipSuccess) rc = rtx_fail(RTX_ERR_HIP, "photon buffer allocation failed");
    }
    if (rc == RTX_OK && hipMalloc((void**)&p->counts, 16 * sizeof(uint32_t)) != hipSuccess) rc = rtx_fail(RTX_ERR_HIP, "photon counter allocation failed");
    if (rc != RTX_OK) {
        rtx_photons_destroy(p);
        return rc;
    }
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipMemset(p->counts, 0, 16 * sizeof(uint32_t));
    (void)hipEventRecord(e0, nullptr);
    launch_photon_emit(rtx_traversal_scene(s), s->cfg, pp->num_photons, pp->first_photon, stride, pp->max_bounces, pp->seed, p->set[0].raw, cap,
                       p->set[1].raw, cap, p->counts);
    (void)hipEventRecord(e1, nullptr);
    uint32_t cnt[4] = {0, 0, 0, 0};
    hipError_t e = hipMemcpy(cnt, p->counts, sizeof(cnt), hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
        rtx_photons_destroy(p);
        return rtx_fail(RTX_ERR_HIP, std::string("photon emission: ") + hipGetErrorString(e));
    }
    float ms = 0.0f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    p->set[0].n = std::min(cnt[0], cap);
    p->set[1].n = std::min(cnt[1], cap);
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->rays_photon = cnt[3];
        stats->total_ms = ms;
    }
    // a strided share is clamped to the whole map's capacity too: what its own smallest keys do not hold, the union's cannot
    rc = clamp_to_capacity(p, 0, pp->num_photons);
    if (rc == RTX_OK) rc = clamp_to_capacity(p, 1, pp->num_photons);
    // PhotonMapBuilder.cpp:38-50: global map scaled by 1/emitted, caustic map by 1/stored
    if (rc == RTX_OK) rc = finalize_set(p, 0, 1.0f / (float)std::max(1, pp->num_photons));
    if (rc == RTX_OK) rc = finalize_set(p, 1, p->set[1].n ? 1.0f / (float)p->set[1].n : 1.0f);
    if (rc != RTX_OK) {
        rtx_photons_destroy(p);
        return rc;
    }
    *out = p;
    return RTX_OK;
}

int rtx_photon_info(const rtx_photons* p, uint64_t counts[2], float radii[2]) {
    if (!p) return rtx_fail(RTX_ERR_INVALID, "photons is NULL");
    if (counts) {
        counts[0] = p->set[0].n;
        counts[1] = p->set[1].n;
    }
    if (radii) {
        radii[0] = p->set[0].radius;
        radii[1] = p->set[1].radius;
    }
    return RTX_OK;
}

int64_t rtx_photon_download(const rtx_photons* p, int which, rtx_photon* out, size_t capacity) {
    if (!p || which < 0 || which > 1) return rtx_fail(RTX_ERR_INVALID, "bad argument");
    const PhotonSet& s = p->set[which];
    const size_t n = std::min<size_t>(capacity, s.n);
    if (!n) return 0;
    if (!out) return rtx_fail(RTX_ERR_INVALID, "out is NULL");
    std::vector<DevPhotonRec> tmp(n);
    if (hipMemcpy(tmp.data(), s.raw, n * sizeof(DevPhotonRec), hipMemcpyDeviceToHost) != hipSuccess) return rtx_fail(RTX_ERR_HIP, "copy failed");
    for (size_t i = 0; i < n; ++i) {
        out[i].pos[0] = tmp[i].pos.x; out[i].pos[1] = tmp[i].pos.y; out[i].pos[2] = tmp[i].pos.z;
        out[i].dir[0] = tmp[i].dir.x; out[i].dir[1] = tmp[i].dir.y; out[i].dir[2] = tmp[i].dir.z;
        out[i].power[0] = tmp[i].power.x; out[i].power[1] = tmp[i].power.y; out[i].power[2] = tmp[i].power.z;
    }
    return (int64_t)n;
}

void* rtx_photon_device_ptr(const rtx_photons* p, int which) {
    if (!p || which < 0 || which > 1) return nullptr;
    return p->set[which].raw;
}

int64_t rtx_photon_copy_raw(const rtx_photons* p, int which, void* dst_dev, size_t capacity) {
    if (!p || which < 0 || which > 1) return rtx_fail(RTX_ERR_INVALID, "bad argument");
    const PhotonSet& s = p->set[which];
    const size_t n = std::min<size_t>(capacity, s.n);
    if (!n) return 0;
    if (!dst_dev) return rtx_fail(RTX_ERR_INVALID, "dst_dev is NULL");
    if (hipMemcpy(dst_dev, s.raw, n * sizeof(DevPhotonRec), hipMemcpyDeviceToDevice) != hipSuccess) return rtx_fail(RTX_ERR_HIP, "copy failed");
    return (int64_t)n;
}

int rtx_photon_set(rtx_photons* p, int which, const void* photons_dev, size_t n, int total_emitted) {
    if (!p || which < 0 || which > 1 || (n && !photons_dev)) return rtx_fail(RTX_ERR_INVALID, "bad argument");
    if (n > 0x7FFFFFFFull) return rtx_fail(RTX_ERR_TOO_LARGE, "too many photons");
    PhotonSet& s = p->set[which];
    if (n > s.capacity) {
        DevPhotonRec* q = nullptr;
        RTX_HIP(hipMalloc((void**)&q, n * sizeof(DevPhotonRec)));
        if (s.raw) (void)hipFree(s.raw);
        s.raw = q;
        s.capacity = (uint32_t)n;
    }
    if (n && photons_dev != s.raw) RTX_HIP(hipMemcpy(s.raw, photons_dev, n * sizeof(DevPhotonRec), hipMemcpyDeviceToDevice));
    s.n = (uint32_t)n;
    const int rc = clamp_to_capacity(p, which, total_emitted);
    if (rc != RTX_OK) return rc;
    const float scale = which == 0 ? 1.0f / (float)std::max(1, total_emitted) : (s.n ? 1.0f / (float)s.n : 1.0f);
    return finalize_set(p, which, scale);
}

int rtx_photon_estimate(rtx_photons* p, int which, const float* positions, const float* normals, size_t n, float* irradiance) {
    if (!p || which < 0 || which > 1 || (n && (!positions || !normals || !irradiance))) return rtx_fail(RTX_ERR_INVALID, "bad argument");
    if (!n) return RTX_OK;
    if (n > 0x7FFFFFFFull) return rtx_fail(RTX_ERR_TOO_LARGE, "more than 2^31 query points in one call");
    DevPhotonMap pm;
    photon_device_map(p, which, pm);
    float *dp = nullptr, *dn = nullptr, *dout = nullptr;
    RTX_HIP(hipMalloc((void**)&dp, n * 12));
    RTX_HIP(hipMalloc((void**)&dn, n * 12));
    RTX_HIP(hipMalloc((void**)&dout, n * 12));
    hipError_t e = hipMemcpy(dp, positions, n * 12, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dn, normals, n * 12, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        launch_photon_probe(pm, dp, dn, (uint32_t)n, dout, nullptr);
        e = hipMemcpy(irradiance, dout, n * 12, hipMemcpyDeviceToHost);
    }
    (void)hipFree(dp);
    (void)hipFree(dn);
    (void)hipFree(dout);
    if (e != hipSuccess) return rtx_fail(RTX_ERR_HIP, std::string("photon estimate: ") + hipGetErrorString(e));
    return RTX_OK;
}

}  // extern "C"
