// C ABI (include/rtx.h): CastRays on the GPU — host orchestration of the wavefront path tracer.
// Replaces RayCaster::CastRays / RenderBucket (reference RayTracer/RayCaster.cpp:697-1023, 1146-1314): buckets of
// this rank -> pixel list -> passes of S samples per pixel through ray-gen / extend / shade / shadow (+ tail kernel) ->
// film accumulation -> confidence-interval test -> refinement rounds.
#include "rtx_internal.h"
#include "pathstate.h"
#include "photon.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

struct PathPoolHost {
    PathPool dev{};
    std::vector<void*> allocs;
    // The arrays of the pool proper are carved from ONE allocation (ensure_pool): slabPhase 1 adds their sizes up, 2 hands the
    // pieces out, 0 (arrays allocated later: legacy state, side spill slab) is a hipMalloc each. With an allocation per array
    // nee_kernel took 35 or 41 ms per C3 frame depending on the process - where the driver happened to put its four streams -
    // and 41 in the first process on a fresh machine; from one allocation it took 37 in 18 of 19 processes.
    char* slab = nullptr;
    size_t slabNeed = 0, slabUsed = 0;
    int slabPhase = 0;
    uint32_t capacity = 0;        // slots of a pass
    uint32_t extra = 0;           // slots of the compaction area behind them (pathstate.h "Compaction")
    int nLights = 0;
    size_t framePixels = 0;
    bool hasCaustic = false;
    std::vector<hipEvent_t> events;
    size_t bytes = 0;
    // second stream: the shadow kernel of bounce k runs beside the extend kernel of bounce k+1
    hipStream_t sideStream = nullptr;
    hipEvent_t sideDone = nullptr;
    hipEvent_t evTotal0 = nullptr, evTotal1 = nullptr;   // whole-call timing; owned here so that no error path leaks them
    uint2* sideSpill = nullptr;   // its own traversal spill slab (both kernels are in flight at once)
    char* legacyState = nullptr;  // in allocs: legacy_kernel's calls and frames (legacy shading mode only)
    // the pixel list of this rank's buckets, kept on the device between calls (building and uploading it costs the host
    // 16 ms per 1080p frame, during which the GPU waits): valid while baseKey = {W, H, tile, rank, world} matches
    int* pixelsBase = nullptr;    // in allocs; never written by the adaptive rounds
    SegQ qCull{};                 // in allocs: control words of the queue over PathPool::cullShadows
    int baseKey[5] = {0, 0, 0, 0, 0};
    uint32_t baseN = 0;
    size_t baseValid = 0;
    void release() {
        if (sideStream) {
            (void)hipStreamSynchronize(sideStream);
            (void)hipStreamDestroy(sideStream);
            sideStream = nullptr;
        }
        if (sideDone) {
            (void)hipEventDestroy(sideDone);
            sideDone = nullptr;
        }
        if (evTotal0) (void)hipEventDestroy(evTotal0);
        if (evTotal1) (void)hipEventDestroy(evTotal1);
        evTotal0 = evTotal1 = nullptr;
        sideSpill = nullptr;   // in allocs
        legacyState = nullptr;
        pixelsBase = nullptr;
        baseKey[0] = 0;
        for (void* p : allocs) (void)hipFree(p);
        allocs.clear();
        for (hipEvent_t e : events) (void)hipEventDestroy(e);
        events.clear();
        capacity = 0;
        extra = 0;
        framePixels = 0;
        bytes = 0;
    }
};

extern "C" void rtx_scene_release_pool(rtx_scene* s) {
    if (s && s->pool) {
        reinterpret_cast<PathPoolHost*>(s->pool)->release();
        delete reinterpret_cast<PathPoolHost*>(s->pool);
        s->pool = nullptr;
    }
}

template <class T>
static int pool_alloc(PathPoolHost* ph, T** out, size_t count) {
    void* p = nullptr;
    size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    if (ph->slabPhase) {
        const size_t piece = (bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);   // 2 MB pieces
        if (ph->slabPhase == 1) {
            ph->slabNeed += piece;
            *out = nullptr;
        } else {
            *out = (T*)(ph->slab + ph->slabUsed);
            ph->slabUsed += piece;
        }
        return RTX_OK;
    }
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return rtx_fail(RTX_ERR_HIP, std::string("path pool hipMalloc: ") + hipGetErrorString(e));
    ph->allocs.push_back(p);
    ph->bytes += bytes;
    *out = (T*)p;
    return RTX_OK;
}

// slots of the compaction area for passes of `slots` paths that are first moved together when 1/div of them are left and again
// whenever 1/step of the last count is left: slots / div * (1 + 1/step + 1/step^2 + ...)
static uint32_t repack_extra(uint32_t slots, int div, int step) {
    if (div <= 0) return 0u;
    return (uint32_t)(((size_t)slots / (size_t)div) * (size_t)step / (size_t)(step - 1)) + 1024u;
}

static int ensure_pool(rtx_scene* s, uint32_t slots, size_t framePixels, size_t listPixels, bool caustic, uint32_t extra) {
    PathPoolHost* ph = reinterpret_cast<PathPoolHost*>(s->pool);
    const int nL = std::max(1, s->dev.n_lights);
    if (ph && ph->capacity >= slots && ph->extra >= extra && ph->framePixels >= std::max(framePixels, listPixels) && ph->nLights == nL &&
        (ph->hasCaustic || !caustic))
        return RTX_OK;
    rtx_scene_release_pool(s);
    ph = new PathPoolHost();
    s->pool = reinterpret_cast<PathPool*>(ph);
    ph->capacity = slots;
    ph->extra = extra;
    ph->nLights = nL;
    ph->framePixels = std::max(framePixels, listPixels);
    ph->hasCaustic = caustic;
    PathPool& d = ph->dev;
    // the arrays a path reads and writes on its way hold the compaction area too; the queues and what belongs to a sample's place
    // in the pass (guides, film position) are laid out for the slots of a pass
    const size_t all = (size_t)slots + extra;
    d.capacity = (uint32_t)all;
    d.baseCap = slots;   // (render_impl sets it for each pass)
    int rc;
    auto layout = [&]() -> int {
#define PA(field, count) \
    if ((rc = pool_alloc(ph, &d.field, (count))) != RTX_OK) return rc;
    PA(rays, 2 * all);
    PA(hits, all);
    PA(L, all);
    PA(beta, all);
    PA(medium, all);
    PA(prevP, all);
    PA(pendBeta, all);
    PA(nee, (size_t)nL * all);
    PA(caustic, caustic ? all : 1);
    PA(origin, extra);
    PA(albedo, slots);
    PA(normal, slots);
    PA(samplePos, slots);
    // queue regions are laid out per producer block of 256 threads: round the capacity up to whole blocks
    const size_t qcap = (((size_t)slots + 255) / 256) * 256;
    PA(queueA, qcap);
    PA(queueB, qcap);
    PA(shadowQ, 2 * ((size_t)nL * qcap + (size_t)QSEG * 256));   // (regions by the blocks of nee_kernel's grid, rounded up to QSEG rows)
    PA(neeCtx, 8 * qcap);   // NEE_CTX_VEC float4 per vertex of a bounce (nee_ctx.h)
    d.neeStride = qcap;
    PA(qA.ctl, SEGQ_WORDS);
    PA(qB.ctl, SEGQ_WORDS);
    PA(qS.ctl, SEGQ_WORDS);
    PA(qN.ctl, SEGQ_WORDS);
    d.qN.run = 4u;   // shade_kernel's waves reserve their pairs one atomic each: interleaved segments (segqueue.h; nee_kernel reads them transposed)
    PA(neeQ, 2 * (size_t)nL * qcap);
    PA(totals, 16);
    PA(cullPaths, RTX_CULL_CAP);
    PA(cullShadows, RTX_CULL_CAP);
    if ((rc = pool_alloc(ph, &ph->qCull.ctl, SEGQ_WORDS)) != RTX_OK) return rc;
    PA(counters, 64);
    // blocker cache of the shadow kernel (variant builds, -DRTX_BLOCKER_CACHE): 2^22 entries (16 MB), cells of 1/512 of the scene's
    // largest extent; RTX_AMD_BLOCKER_CACHE=0 switches it off, =N sets the cells per extent
    {
        int cells = 0;   // off: the shipped shadow kernel does not read it (kernels_path.hip, blocker_key)
#ifdef RTX_BLOCKER_CACHE
        cells = 512;
        if (const char* e = std::getenv("RTX_AMD_BLOCKER_CACHE")) cells = std::atoi(e);
#endif
        d.blockerCache = nullptr;
        if (cells > 0 && !s->dev.has_xf) {
            const float ext = std::max(s->dev.bounds[3] - s->dev.bounds[0], std::max(s->dev.bounds[4] - s->dev.bounds[1], s->dev.bounds[5] - s->dev.bounds[2]));
            if (ext > 0.0f && ext < 1e30f) {
                d.blockerMask = (1u << 22) - 1u;
                d.blockerScale = (float)cells / ext;
                PA(blockerCache, (size_t)d.blockerMask + 1);
                if (ph->slabPhase != 1 && hipMemset(d.blockerCache, 0xFF, ((size_t)d.blockerMask + 1) * sizeof(int)) != hipSuccess) return rtx_fail(RTX_ERR_HIP, "blocker cache clear failed");
            }
        }
    }
    PA(pixels, ph->framePixels + 64);
    PA(pixelsNext, ph->framePixels + 64);
    if ((rc = pool_alloc(ph, &ph->pixelsBase, ph->framePixels + 64)) != RTX_OK) return rc;
    PA(statSum, ph->framePixels);
    PA(statSum2, ph->framePixels);
    PA(rayCounters, 8);
#undef PA
    return RTX_OK;
    };
    ph->slabPhase = 1;
    ph->slabNeed = 0;
    rc = layout();
    if (rc == RTX_OK) {
        void* p = nullptr;
        const char* sw = std::getenv("RTX_AMD_POOL_SLAB");   // =0: an allocation per array (the fallback below, for the tests)
        hipError_t e = (sw && std::atoi(sw) == 0) ? hipErrorOutOfMemory : hipMalloc(&p, ph->slabNeed);
        if (e == hipSuccess) {
            ph->allocs.push_back(p);
            ph->bytes += ph->slabNeed;
            ph->slab = (char*)p;
            ph->slabUsed = 0;
            ph->slabPhase = 2;
        } else {
            (void)hipGetLastError();   // no piece of that size: an allocation per array, as before round 3
            ph->slabPhase = 0;
        }
        rc = layout();
    }
    ph->slabPhase = 0;
    return rc;
}

// Pixel list of the buckets owned by (rank, world): RayCaster.cpp:1152-1158 builds 16x16 buckets row-major; bucket b
// belongs to rank b % world. Inside a bucket pixels are emitted as 16x4 blocks (one wave each).
static void build_pixel_list(int W, int H, int bs, int rank, int world, std::vector<int>& list, size_t& nValid) {
    list.clear();
    nValid = 0;
    int b = 0;
    for (int y = 0; y < H; y += bs)
        for (int x = 0; x < W; x += bs, ++b) {
            if (b % world != rank) continue;
            const int x1 = std::min(x + bs, W), y1 = std::min(y + bs, H);
            // blocks of up to 16 columns x 4 rows -> exactly 64 entries each (pad with -1)
            for (int by = y; by < y1; by += 4)
                for (int bx = x; bx < x1; bx += 16) {
                    for (int r = 0; r < 4; ++r)
                        for (int c = 0; c < 16; ++c) {
                            const int px = bx + c, py = by + r;
                            if (px < x1 && py < y1) {
                                list.push_back(py * W + px);
                                ++nValid;
                            } else
                                list.push_back(-1);
                        }
                }
        }
}

// Host-only view of the tile sharding (no device needed): the padded pixel list of (rank, world).
extern "C" int64_t rtx_debug_pixel_list(int W, int H, int tile, int rank, int world, int* out, size_t capacity) {
    if (W <= 0 || H <= 0 || world < 1 || rank < 0 || rank >= world) return rtx_fail(RTX_ERR_INVALID, "bad argument");
    std::vector<int> list;
    size_t nValid = 0;
    build_pixel_list(W, H, tile > 0 ? tile : 16, rank, world, list, nValid);
    if (out) std::memcpy(out, list.data(), std::min(capacity, list.size()) * sizeof(int));
    return (int64_t)list.size();
}

struct PassTimers {
    PathPoolHost* ph;
    size_t used = 0;
    std::vector<std::pair<int, std::pair<size_t, size_t>>> spans;   // kind, (start,stop) event index
    std::vector<uint32_t> items;   // work items of each span (RTX_AMD_TRACE)
    bool trace = std::getenv("RTX_AMD_TRACE") != nullptr;
    uint32_t nextItems = 0;
    double ms[4] = {0, 0, 0, 0};   // extend, shadow, shade + tail, nee
    hipEvent_t get() {
        if (used == ph->events.size()) {
            hipEvent_t e;
            (void)hipEventCreate(&e);
            ph->events.push_back(e);
        }
        return ph->events[used++];
    }
    void begin(int kind, hipStream_t st) {
        hipEvent_t e = get();
        (void)hipEventRecord(e, st);
        spans.push_back({kind, {used - 1, 0}});
        if (trace) items.push_back(nextItems);
    }
    void end(hipStream_t st) {
        hipEvent_t e = get();
        (void)hipEventRecord(e, st);
        spans.back().second.second = used - 1;
    }
    void flush() {   // caller has synchronised the stream
        size_t k = 0;
        for (auto& s : spans) {
            float t = 0.0f;
            if (hipEventElapsedTime(&t, ph->events[s.second.first], ph->events[s.second.second]) == hipSuccess) ms[s.first] += t;
            if (trace) {   // development aid: one line per timed launch
                static const char* names[4] = {"extend", "shadow", "shade/tail", "nee"};
                std::fprintf(stderr, "[rtx trace] %-10s %10u items %9.3f ms\n", names[s.first], items[k++], t);
            }
        }
        items.clear();
        spans.clear();
        used = 0;
    }
};

// One explicit round on a caller-given pixel list (rtx_render_round): samples [sampleBase, sampleBase + nSamples) of
// every listed pixel, per-pixel statistics kept in caller-owned device buffers, no confidence-interval step.
struct RoundSpec {
    const int* pixels;
    size_t nPixels;
    int sampleBase, nSamples, resetZ;
    float4* statSum;
    float4* statSum2;
};

static int render_impl(rtx_scene* s, const rtx_camera* cam, const rtx_render_params* rp, rtx_photons* photons, const rtx_film* film,
                       rtx_stats* stats, float4* samplesOut, const RoundSpec* spec = nullptr) {
    if (!s || !cam || !rp || !film) return rtx_fail(RTX_ERR_INVALID, "NULL argument");
    if (!film->accum || !film->count || !film->z) return rtx_fail(RTX_ERR_INVALID, "film.accum/count/z must be device buffers");
    if (!spec && (rp->spp_min < 1 || rp->spp_max < rp->spp_min)) return rtx_fail(RTX_ERR_INVALID, "need 1 <= spp_min <= spp_max");
    if (spec && (spec->nSamples < 1 || spec->sampleBase < 0)) return rtx_fail(RTX_ERR_INVALID, "need n_samples >= 1, sample_base >= 0");
    if (rp->max_bounces < 1) return rtx_fail(RTX_ERR_INVALID, "max_bounces must be >= 1");
    if (rp->shading_mode != RTX_SHADING_PATH && rp->shading_mode != RTX_SHADING_LEGACY)
        return rtx_fail(RTX_ERR_INVALID, "unknown shading_mode");
    if (rp->filter_type < RTX_FILTER_TENT || rp->filter_type > RTX_FILTER_LANCZOS) return rtx_fail(RTX_ERR_INVALID, "unknown filter_type");
    // every sample splats (2r+1)^2 pixels: an unbounded radius would turn the film kernel into a frame-sized loop per sample
    if (!(rp->filter_radius <= RTX_MAX_FILTER_RADIUS)) return rtx_fail(RTX_ERR_INVALID, "filter_radius must be <= RTX_MAX_FILTER_RADIUS (<= 0: the filter's default support)");
    const int world = rp->world > 0 ? rp->world : 1;
    if (rp->rank < 0 || rp->rank >= world) return rtx_fail(RTX_ERR_INVALID, "rank out of range");
    DevCamera dc;
    int rc;
    if ((rc = make_dev_camera(cam, dc)) != RTX_OK) return rc;
    const int W = dc.W, H = dc.H;
    const int stride = rp->index_stride > 0 ? rp->index_stride : 1024;
    const int sppMax = spec ? spec->sampleBase + spec->nSamples : rp->spp_max;
    const int perRound = spec ? spec->nSamples : rp->spp_min;
    if ((long long)W * H * stride + sppMax + 1 > 2147483647LL)
        return rtx_fail(RTX_ERR_TOO_LARGE, "sampleIndex = (y*W+x)*index_stride+s+1 overflows int (RayCaster.cpp:784)");
    if (sppMax > stride) return rtx_fail(RTX_ERR_INVALID, "spp_max exceeds index_stride");
    const int bs = rp->tile_size > 0 ? rp->tile_size : 16;

    std::vector<int> list;
    size_t nValid = 0;
    if (spec) {
        if (spec->nPixels && !spec->pixels) return rtx_fail(RTX_ERR_INVALID, "pixel list is NULL");
        list.assign(spec->pixels, spec->pixels + spec->nPixels);
        for (int p : list)
            if (p < 0 || p >= W * H) return rtx_fail(RTX_ERR_INVALID, "pixel index outside the frame");
        nValid = list.size();
        list.resize((list.size() + 63) & ~(size_t)63, -1);   // whole waves
    }
    const int listKey[5] = {W, H, bs, rp->rank, world};
    auto list_cached = [&](const PathPoolHost* h) { return h && h->pixelsBase && std::memcmp(h->baseKey, listKey, sizeof(listKey)) == 0; };
    bool cached = !spec && list_cached(reinterpret_cast<PathPoolHost*>(s->pool));
    if (cached) {
        nValid = reinterpret_cast<PathPoolHost*>(s->pool)->baseValid;
    } else if (!spec)
        build_pixel_list(W, H, bs, rp->rank, world, list, nValid);
    uint32_t nPix = cached ? reinterpret_cast<PathPoolHost*>(s->pool)->baseN : (uint32_t)list.size();

    DevPhotonMap pm;
    std::memset(&pm, 0, sizeof(pm));
    const bool usePhotons = photons && rp->use_photons;
    if (usePhotons) photon_device_map(photons, 1, pm);
    const bool legacy = rp->shading_mode == RTX_SHADING_LEGACY;
    DevPhotonMap pmGlobal;
    std::memset(&pmGlobal, 0, sizeof(pmGlobal));
    if (usePhotons && legacy) photon_device_map(photons, 0, pmGlobal);   // gathered in legacy mode only (Shading_Legacy.cpp:440-444)

    // Compaction of the path state inside a pass (pathstate.h): the bounce that first finds at most 1/repackDiv of the pass's
    // paths alive writes their state to consecutive slots behind the pass's own, and so does every later bounce that finds
    // 1/repackStep of the last such count; bounces of fewer than repackMin paths are left alone (the tail kernel is about to
    // take them). The move itself is free - the bounce writes its paths' state anyway - and the area costs slots / 4 of the
    // per-path arrays (C3: 13 GB). Measured on the C3 frame: 777.5 ms without, 766 / 760 / 755 ms with (8, 4) / (6, 3) / (8, 2).
    // Results do not depend on it. RTX_AMD_REPACK=0 switches it off; RTX_AMD_REPACK_DIV / _STEP / _MIN set the three numbers
    // (tests force a move at the first bounce of tiny passes with DIV=1, MIN=0).
    int repackDiv = legacy ? 0 : 8;
    uint32_t repackMin = 4u * RTX_TAIL_MAX_PATHS;
    if (const char* e = std::getenv("RTX_AMD_REPACK")) repackDiv = std::atoi(e) != 0 ? repackDiv : 0;
    if (const char* e = std::getenv("RTX_AMD_REPACK_DIV")) repackDiv = repackDiv > 0 ? std::max(1, std::atoi(e)) : 0;
    if (const char* e = std::getenv("RTX_AMD_REPACK_MIN")) repackMin = (uint32_t)std::atol(e);
    int repackStep = 2;
    if (const char* e = std::getenv("RTX_AMD_REPACK_STEP")) repackStep = std::max(2, std::atoi(e));

    // path pool sizing: S samples per pixel per pass
    int S = rp->samples_per_pass;
    if (S <= 0) {
        // Size the path pool for the 288 GB of HBM: the deeper bounces of a pass hold few paths (Russian roulette),
        // so fewer, larger passes amortise that tail. Use up to 60 % of the free memory, at most 170 GB.
        size_t freeB = 0, totalB = 0;
        if (hipMemGetInfo(&freeB, &totalB) != hipSuccess) freeB = (size_t)8 << 30;
        size_t perSlot = 192 + 128 + 80 * (size_t)std::max(1, s->dev.n_lights);   // + NEE record, + 32 B per light of NEE queue
        if (repackDiv > 0) perSlot += (148 + 16 * (size_t)std::max(1, s->dev.n_lights)) * (size_t)repackStep / (size_t)(repackStep - 1) / (size_t)repackDiv + 1;   // compaction area
        const size_t budget = std::min<size_t>((size_t)(0.6 * (double)freeB), (size_t)170 << 30);
        const size_t target = std::max<size_t>(budget / perSlot, 1u << 20);
        S = (int)std::max<size_t>(1, std::min<size_t>((size_t)perRound, target / std::max<uint32_t>(nPix, 1)));
        // a pool kept from an earlier call is memory already paid for (and no longer "free"): use all of it
        if (s->pool && nPix) {
            const uint32_t have = reinterpret_cast<PathPoolHost*>(s->pool)->capacity / nPix;
            S = std::max(S, (int)std::min<uint32_t>(have, (uint32_t)perRound));
        }
    }
    S = std::min(S, perRound);
    // slot * 8 + light must fit 32 bits (shadow queue entries), for the slots of the compaction area too
    auto all_slots = [&](int Sx) { return (size_t)nPix * Sx + repack_extra((uint32_t)std::min<size_t>((size_t)nPix * Sx, 0xFFFFFFFFull), repackDiv, repackStep); };
    while (S > 1 && all_slots(S) > 0x1FFFFFFFull) --S;
    if (all_slots(S) > 0x1FFFFFFFull) return rtx_fail(RTX_ERR_TOO_LARGE, "more than 2^29 path slots per pass");
    // shadow queue entries are indexed with 32 bits
    while (S > 1 && ((size_t)nPix * S + 256) * (size_t)std::max(1, s->dev.n_lights) > 0xFFFFFFFFull) --S;
    if (((size_t)nPix * S + 256) * (size_t)std::max(1, s->dev.n_lights) > 0xFFFFFFFFull)
        return rtx_fail(RTX_ERR_TOO_LARGE, "more than 2^32 shadow segments per pass");
    if ((rc = ensure_pool(s, nPix * (uint32_t)S, (size_t)W * H, nPix, usePhotons, repack_extra(nPix * (uint32_t)S, repackDiv, repackStep))) != RTX_OK) return rc;
    PathPoolHost* ph = reinterpret_cast<PathPoolHost*>(s->pool);
    PathPool pool = ph->dev;
    if (spec) {
        pool.statSum = spec->statSum;
        pool.statSum2 = spec->statSum2;
    }
    hipStream_t st = s->cfg.stream;

    rtx_stats out;
    std::memset(&out, 0, sizeof(out));
    if (!ph->evTotal0) RTX_HIP(hipEventCreate(&ph->evTotal0));
    if (!ph->evTotal1) RTX_HIP(hipEventCreate(&ph->evTotal1));
    const hipEvent_t evTotal0 = ph->evTotal0, evTotal1 = ph->evTotal1;
    RTX_HIP(hipEventRecord(evTotal0, st));

    if (!spec && !list_cached(ph)) {   // first call with this frame / sharding, or the pool was rebuilt
        if (list.empty()) build_pixel_list(W, H, bs, rp->rank, world, list, nValid);
        if (nPix) RTX_HIP(hipMemcpyAsync(ph->pixelsBase, list.data(), (size_t)nPix * sizeof(int), hipMemcpyHostToDevice, st));
        RTX_HIP(hipStreamSynchronize(st));   // `list` is pageable host memory that goes out of scope with this call
        std::memcpy(ph->baseKey, listKey, sizeof(listKey));
        ph->baseN = nPix;
        ph->baseValid = nValid;
    } else if (spec && nPix)
        RTX_HIP(hipMemcpyAsync(pool.pixels, list.data(), (size_t)nPix * sizeof(int), hipMemcpyHostToDevice, st));
    if (!spec) launch_stats_clear(pool, (uint32_t)((size_t)W * H), st);
    RTX_HIP(hipMemsetAsync(pool.rayCounters, 0, 8 * sizeof(unsigned long long), st));
    RTX_HIP(hipMemsetAsync(pool.totals, 0, 16 * sizeof(uint32_t), st));
    RTX_HIP(hipMemsetAsync(ph->qCull.ctl, 0, SEGQ_WORDS * sizeof(uint32_t), st));
    if (s->cfg.counting) RTX_HIP(hipMemsetAsync(s->cfg.counters, 0, 8 * sizeof(unsigned long long), st));
    // queue counters are left at zero by every scan; a cancelled or failed call may have left some behind
    RTX_HIP(hipMemsetAsync(pool.qA.ctl, 0, SEGQ_WORDS * sizeof(uint32_t), st));
    RTX_HIP(hipMemsetAsync(pool.qB.ctl, 0, SEGQ_WORDS * sizeof(uint32_t), st));
    RTX_HIP(hipMemsetAsync(pool.qS.ctl, 0, SEGQ_WORDS * sizeof(uint32_t), st));
    RTX_HIP(hipMemsetAsync(pool.qN.ctl, 0, SEGQ_WORDS * sizeof(uint32_t), st));
    const uint32_t shadowMult = (uint32_t)std::max(1, s->dev.n_lights);
    uint32_t tailMax = RTX_TAIL_MAX_PATHS;
    if (const char* e = std::getenv("RTX_AMD_TAIL_PATHS")) tailMax = std::min<uint32_t>(tailMax, (uint32_t)std::atol(e));   // 0 disables
    if (s->cfg.counting) tailMax = 0;   // node / triangle tallies exist in the counting build of extend_kernel only

    // Overlap: shadow(k) on a side stream beside extend(k+1); shade(k+1) waits for both. Either kernel
    // alone leaves most of the chip idle while its last rays finish (a launch costs ~0.3 ms beyond its throughput time).
    // Only for bounces of at most overlapMax paths: larger launches fill the chip on their own, and sharing it would
    // stretch the extend launches that the roofline figure is measured on.
    int shadowFlags = 0;   // bit 1: shadow hops run to their closest hit even when an earlier hit decides the chain
    if (const char* e = std::getenv("RTX_AMD_SHADOW_EARLY_EXIT")) shadowFlags = std::atoi(e) != 0 ? 0 : 2;
    bool overlap = true;
    uint32_t overlapMax = 2u << 20;
    if (const char* e = std::getenv("RTX_AMD_OVERLAP")) overlap = std::atoi(e) != 0;
    if (const char* e = std::getenv("RTX_AMD_OVERLAP_MAX")) overlapMax = (uint32_t)std::atol(e);
    if (overlap && !legacy && !ph->sideStream) {
        RTX_HIP(hipStreamCreateWithFlags(&ph->sideStream, hipStreamNonBlocking));
        RTX_HIP(hipEventCreateWithFlags(&ph->sideDone, hipEventDisableTiming));
        if ((rc = pool_alloc(ph, &ph->sideSpill, (size_t)s->cfg.gridBlocks * TRAV_BLOCK * STACK_GLOBAL)) != RTX_OK) {
            rtx_scene_release_pool(s);   // stream, event and every pool allocation; the next call starts from scratch
            return rc;
        }
    }
    LaunchCfg sideCfg = s->cfg;
    sideCfg.stream = ph->sideStream;
    sideCfg.spill = ph->sideSpill;
    bool sidePending = false;   // a shadow launch on the side stream that the main stream has not waited for yet
    const int shadeKinds = (s->hasMicrofacet ? 1 : 0) | (s->hasPhong ? 2 : 0);   // which forms of the shade kernel the scene needs
    // Scenes of world-space triangles run the lean kernels, which leave out the reference's root and object box tests
    // (traverse.h ref_box_hit); with `culls` the shade and shadow kernels find the hits those tests would have removed - about
    // 2 in 10^7 samples of the C3 frame - and the general kernels trace them again: the vertex through the tail kernel, the
    // shadow chain through a second, tiny shadow launch. On by default since round 3 - every sample of the path loop is then
    // the oracle's bit for bit - at 4 % of the C3 frame (shadow 162 -> 186 ms: an object id and a box to fetch at the end of
    // every blocked hop; the general kernels would cost 20 %). RTX_AMD_BOX_CULLS=0 switches the checks off.
    bool culls = !legacy && !s->dev.has_xf && !s->cfg.counting;
    if (const char* e = std::getenv("RTX_AMD_BOX_CULLS")) culls = culls && std::atoi(e) != 0;
    if (!culls) pool.cullPaths = pool.cullShadows = nullptr;
    DevScene devGeneral = s->dev;
    if (culls) devGeneral.has_xf = 1;   // tail kernel and re-traces: the instantiation that repeats the box tests
    s->cancel.store(0);
    PassTimers tm{ph};
    int* curPixels = spec ? pool.pixels : ph->pixelsBase;
    int* nextPixels = spec ? pool.pixelsNext : pool.pixels;
    int* sparePixels = pool.pixelsNext;   // takes the place of pixelsBase in the rotation: the base list is never overwritten
    size_t curValid = nValid;
    int roundBase = spec ? spec->sampleBase : 0;
    for (int round = 0; nPix > 0 && curValid > 0; ++round) {
        const int target = std::min(perRound, sppMax - roundBase);
        if (target <= 0) break;
        for (int done = 0; done < target && !s->cancel.load(std::memory_order_relaxed);) {
            const int Sp = std::min(S, target - done);
            const int sBase = roundBase + done;
            // ---- one wavefront pass ----
            uint32_t* qin = pool.queueA;
            uint32_t* qout = pool.queueB;
            SegQ sqIn = pool.qA, sqOut = pool.qB;
            const uint32_t gridGen = launch_raygen(dc, pool, curPixels, nPix, Sp, sBase, stride, sqIn, qin, legacy ? 1 : 0, st);
            launch_segq_scan(sqIn, gridGen, 256, 1, pool.totals, 2, st);
            uint32_t n = (uint32_t)(curValid * (size_t)Sp);   // = totals[2]: every valid pixel generates Sp paths
            out.paths += n;
            pool.baseCap = nPix * (uint32_t)Sp;   // the compaction area starts behind this pass's slots
            uint32_t extraUsed = 0, lastRepackN = 0;
            if (legacy && n > 0 && !ph->legacyState) {
                if ((rc = pool_alloc(ph, &ph->legacyState, legacy_state_bytes(s->cfg))) != RTX_OK) {
                    rtx_scene_release_pool(s);
                    return rc;
                }
            }
            if (legacy && n > 0) {
                // primary hits by the extend kernel, then the whole recursion of every sample in one kernel
                out.rays_primary += n;
                tm.begin(0, st);
                launch_extend_queue(rtx_traversal_scene(s), s->cfg, pool.rays, sqIn, qin, n, RTX_HIT_FRONT_AND_BACK, 1, pool.hits);
                tm.end(st);
                out.extend_launches++;
                tm.begin(2, st);
                launch_legacy(rtx_traversal_scene(s), s->cfg, dc, pool, pmGlobal, pm, sqIn, qin, n, rp->seed, sBase, Sp, curPixels,
                              (!s->hasMicrofacet && !s->hasPhong) ? 1 : 0, ph->legacyState);
                tm.end(st);
                RTX_HIP(hipStreamSynchronize(st));   // one pass at a time, so that rtx_cancel is seen between passes
                n = 0;
            }
            for (int bounce = 0; bounce < rp->max_bounces && n > 0 && !s->cancel.load(std::memory_order_relaxed); ++bounce) {
                if (bounce == 0) out.rays_primary += n;
                else out.rays_bounce += n;
                tm.nextItems = n;
                tm.begin(0, st);
                launch_extend_queue(s->dev, s->cfg, pool.rays, sqIn, qin, n, RTX_HIT_FRONT_AND_BACK, 1, pool.hits);
                tm.end(st);
                out.extend_launches++;
                if (sidePending) {
                    RTX_HIP(hipStreamWaitEvent(st, ph->sideDone, 0));
                    sidePending = false;
                }
                // shade: the vertex itself (hit, bump, emission, BSDF sample, next ray); nee: its light samples, one lane per
                // (vertex, light), which appends the shadow segments (kernels_nee.hip)
                uint32_t repackBase = 0u;
                if (repackDiv > 0 && n > repackMin && n <= ph->extra - extraUsed && n <= (lastRepackN ? lastRepackN / (uint32_t)repackStep : pool.baseCap / (uint32_t)repackDiv)) {
                    repackBase = pool.baseCap + extraUsed;   // this bounce writes its paths to consecutive slots from here on
                    extraUsed += n;
                    lastRepackN = n;
                    out.repacks++;
                }
                tm.begin(2, st);
                const uint32_t gridShade = launch_shade(s->dev, dc, pool, pm, sqIn, qin, n, bounce, rp->max_bounces, rp->seed, sBase, Sp,
                                                        curPixels, sqOut, qout, pool.qS, shadeKinds, st, repackBase);
                tm.end(st);
                launch_segq_scan2(sqOut, gridShade, 1, pool.qN, gridShade, shadowMult, RTX_SHADE_BLOCK, pool.totals, st);   // totals[0] paths alive, [1] pairs
                // The number of pairs, read back for the nee grid: sized by its upper bound (vertices x lights, three quarters of it
                // blocks that find nothing and leave) the launch costs 7 ms more per C3 frame than this second round trip per bounce.
                uint32_t neeBound = 0;
                {
                    uint32_t pc[2] = {0, 0};
                    RTX_HIP(hipMemcpyAsync(pc, pool.totals, sizeof(pc), hipMemcpyDeviceToHost, st));
                    RTX_HIP(hipStreamSynchronize(st));
                    neeBound = pc[1] ? pc[1] : 1u;
                }
                tm.begin(3, st);
                const uint32_t gridNee = launch_nee(s->dev, pool, n, pool.qS, shadeKinds, st, neeBound);
                tm.end(st);
                out.shade_launches++;
                out.vertices += n;
                // (the shadow queue is laid out by the blocks of the nee grid that held pairs: their number comes from the pair count)
                launch_segq_scan(pool.qS, gridNee, 256, 1, pool.totals, 1, st, pool.qN.ctl + SEGQ_PREFIX + QSEG);   // totals[1] shadow segments
                uint32_t cnt[6] = {0, 0, 0, 0, 0, 0};   // paths alive, shadow segments, -, -, culled vertices, culled chains
                RTX_HIP(hipMemcpyAsync(cnt, pool.totals, sizeof(cnt), hipMemcpyDeviceToHost, st));
                RTX_HIP(hipStreamSynchronize(st));
                const bool side = overlap && cnt[1] > 0 && cnt[0] > 0 && cnt[0] <= overlapMax;   // cnt[0] = 0: nothing to overlap with
                hipStream_t sst = side ? ph->sideStream : st;
                if (cnt[1]) {
                    tm.nextItems = cnt[1];
                    tm.begin(1, sst);
                    launch_shadow(s->dev, side ? sideCfg : s->cfg, pool, pool.qS, cnt[1], (bounce <= 8 ? 1 : 0) | shadowFlags);   // useTransmittance (RayCaster.cpp:454)
                    // (the chains the lean kernel set aside - a few dozen per frame - through the general kernel: two blocks that follow
                    // one ray at a time, 90 us when a bounce has any; moving them to the side stream, beside the next bounce's extend
                    // launch, was measured and changes nothing: 668 against 667 ms per C3 frame)
                    if (culls) launch_shadow_culled(devGeneral, side ? sideCfg : s->cfg, pool, ph->qCull, (bounce <= 8 ? 1 : 0) | shadowFlags);
                    tm.end(sst);
                    out.shadow_launches++;
                    out.shadow_segments += cnt[1];
                }
                if (side) {
                    RTX_HIP(hipEventRecord(ph->sideDone, ph->sideStream));
                    sidePending = true;
                }
                if (culls && cnt[4] > 0) {
                    // vertices of this bounce whose hit the reference would not have reached: their paths go on in the general
                    // tail kernel, from this bounce's ray (still in pool.rays) to their ends
                    const uint32_t nC = std::min<uint32_t>(cnt[4], RTX_CULL_CAP);
                    // (the transmittances of the previous bounce, which the tail kernel's resolve reads, were waited for in front
                    // of this bounce's shade launch; this bounce's shadow launch holds no segment of these paths)
                    launch_segq_identity(s->cfg.identQ, nC, st);
                    tm.nextItems = nC;
                    tm.begin(2, st);
                    launch_tail(devGeneral, s->cfg, dc, pool, pm, s->cfg.identQ, pool.cullPaths, nC, bounce, rp->max_bounces, rp->seed, sBase, Sp, curPixels,
                                (s->hasMicrofacet || s->hasPhong) ? 1 : 0, pm.n > 0 ? 1 : 0);
                    tm.end(st);
                    RTX_HIP(hipMemsetAsync(pool.totals + 4, 0, sizeof(uint32_t), st));
                    if (bounce == 0) out.rays_primary -= nC;   // traced again (and counted) by the tail kernel
                    else out.rays_bounce -= nC;
                }
                std::swap(qin, qout);
                std::swap(sqIn, sqOut);
                n = cnt[0];
                if (n > 0 && n <= tailMax && bounce + 1 < rp->max_bounces) {
                    // few paths left: one kernel walks them to their ends (tail_kernel)
                    if (sidePending) {
                        RTX_HIP(hipStreamWaitEvent(st, ph->sideDone, 0));
                        sidePending = false;
                    }
                    tm.nextItems = n;
                    tm.begin(2, st);
                    launch_tail(devGeneral, s->cfg, dc, pool, pm, sqIn, qin, n, bounce + 1, rp->max_bounces, rp->seed, sBase, Sp, curPixels,
                                (s->hasMicrofacet || s->hasPhong) ? 1 : 0, pm.n > 0 ? 1 : 0);
                    tm.end(st);
                    n = 0;
                }
                if (tm.spans.size() > 2048) {
                    RTX_HIP(hipStreamSynchronize(st));
                    if (ph->sideStream) RTX_HIP(hipStreamSynchronize(ph->sideStream));
                    tm.flush();
                }
            }
            if (sidePending) {   // a cancelled pass, or max_bounces reached with paths alive
                RTX_HIP(hipStreamWaitEvent(st, ph->sideDone, 0));
                sidePending = false;
            }
            // a pass interrupted by rtx_cancel holds unfinished paths: it is dropped, earlier passes stay
            if (!s->cancel.load(std::memory_order_relaxed) && extraUsed)
                launch_unpack(pool, extraUsed, s->dev.n_lights, pm.n > 0 ? 1 : 0, st);   // results of the moved paths back to their samples' slots
            if (!s->cancel.load(std::memory_order_relaxed))
                launch_film(pool, *film, curPixels, nPix, Sp, W, H, rp->filter_type, rp->filter_radius,
                            (done == 0 && (spec ? spec->resetZ != 0 : round == 1)) ? 1 : 0, samplesOut, sBase, sppMax,
                            legacy ? 3.0e38f : 10.0f, s->dev.n_lights, pm.n > 0 ? 1 : 0, st);
            done += Sp;
        }
        roundBase += target;
        if (s->cancel.load(std::memory_order_relaxed)) break;
        if (spec || rp->spp_max <= rp->spp_min || roundBase >= rp->spp_max) break;
        // ---- confidence-interval test -> next pixel list ----
        launch_fill_int(nextPixels, nPix, -1, st);
        RTX_HIP(hipMemsetAsync(pool.counters + 8, 0, sizeof(uint32_t), st));
        launch_adaptive(pool, curPixels, nPix, rp->ci_halfwidth, rp->spp_max, nextPixels, pool.counters + 8, st);
        uint32_t cnt = 0;
        RTX_HIP(hipMemcpyAsync(&cnt, pool.counters + 8, sizeof(cnt), hipMemcpyDeviceToHost, st));
        RTX_HIP(hipStreamSynchronize(st));
        curValid = cnt;
        nPix = (cnt + 63u) & ~63u;
        {
            int* const done = curPixels;
            curPixels = nextPixels;
            nextPixels = (done == ph->pixelsBase) ? sparePixels : done;
        }
    }
    RTX_HIP(hipEventRecord(evTotal1, st));
    RTX_HIP(hipStreamSynchronize(st));
    RTX_HIP(hipGetLastError());
    tm.flush();
    float total = 0.0f;
    (void)hipEventElapsedTime(&total, evTotal0, evTotal1);
    unsigned long long rc2[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // [0] shadow hops, [1] closest-hit rays traced inside tail_kernel, [4] blocker-cache hits (SHADOW_CACHE_STATS builds)
    RTX_HIP(hipMemcpy(rc2, pool.rayCounters, sizeof(rc2), hipMemcpyDeviceToHost));
    out.rays_shadow = rc2[0];
    if (tm.trace && rc2[4]) std::fprintf(stderr, "[rtx trace] blocker cache: %llu of %llu shadow hops started from a cached blocker\n", rc2[4], rc2[0]);
    out.rays_extend = out.rays_primary + out.rays_bounce;   // traced by extend launches (the rest: tail / legacy kernel)
    out.rays_bounce += rc2[1];
    out.extend_ms = tm.ms[0];
    out.shadow_ms = tm.ms[1];
    out.shade_ms = tm.ms[2] + tm.ms[3];
    out.nee_ms = tm.ms[3];
    out.total_ms = total;
    if (s->cfg.counting) {
        unsigned long long c[5];
        RTX_HIP(hipMemcpy(c, s->cfg.counters, sizeof(c), hipMemcpyDeviceToHost));
        out.nodes_visited = c[0];
        out.tris_tested = c[1];
        out.shadow_nodes = c[3];
        out.shadow_tris = c[4];
    }
    if (stats) *stats = out;
    return RTX_OK;
}

extern "C" {

int rtx_cancel(rtx_scene* s) {
    if (!s) return rtx_fail(RTX_ERR_INVALID, "scene is NULL");
    s->cancel.store(1);
    return RTX_OK;
}

int rtx_render_round(rtx_scene* s, const rtx_camera* cam, const rtx_render_params* rp, rtx_photons* photons, const rtx_film* film,
                     const rtx_pixel_stats* st, const int* pixels, size_t n_pixels, int sample_base, int n_samples, int reset_z,
                     rtx_stats* stats) {
    if (!st || !st->sum || !st->sum2) return rtx_fail(RTX_ERR_INVALID, "pixel statistics buffers are NULL");
    RoundSpec spec{pixels, n_pixels, sample_base, n_samples, reset_z, (float4*)st->sum, (float4*)st->sum2};
    return render_impl(s, cam, rp, photons, film, stats, nullptr, &spec);
}

int64_t rtx_refine_list(const rtx_pixel_stats* st, int width, int height, float ci_halfwidth, int spp_max, int* out_pixels,
                        size_t capacity) {
    if (!st || !st->sum || !st->sum2 || width <= 0 || height <= 0) return rtx_fail(RTX_ERR_INVALID, "bad argument");
    if (rtx_device_count() <= 0) return rtx_fail(RTX_ERR_NO_DEVICE, "no HIP device present");
    const uint32_t n = (uint32_t)((size_t)width * height);
    // list + counter in one buffer that is kept between calls (an adaptive frame asks once per round)
    static std::mutex mu;
    static DevBuf scratch[64];
    int dev = 0;
    RTX_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return rtx_fail(RTX_ERR_INVALID, "device index out of range");
    std::lock_guard<std::mutex> lock(mu);
    int rc = scratch[dev].ensure((size_t)n * sizeof(int) + 256);
    if (rc != RTX_OK) return rc;
    uint32_t* dCount = (uint32_t*)scratch[dev].p;
    int* dList = (int*)((char*)scratch[dev].p + 256);
    RTX_HIP(hipMemset(dCount, 0, sizeof(uint32_t)));
    PathPool view{};
    view.statSum = (float4*)st->sum;
    view.statSum2 = (float4*)st->sum2;
    launch_adaptive(view, nullptr, n, ci_halfwidth, spp_max, dList, dCount, nullptr);
    uint32_t cnt = 0;
    hipError_t e = hipMemcpy(&cnt, dCount, sizeof(cnt), hipMemcpyDeviceToHost);
    std::vector<int> host(cnt);
    if (e == hipSuccess && cnt) e = hipMemcpy(host.data(), dList, (size_t)cnt * sizeof(int), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return rtx_fail(RTX_ERR_HIP, std::string("refine list: ") + hipGetErrorString(e));
    std::sort(host.begin(), host.end());   // the compaction order depends on wave scheduling; ranks must agree
    if (out_pixels) std::memcpy(out_pixels, host.data(), std::min(capacity, (size_t)cnt) * sizeof(int));
    return (int64_t)cnt;
}

int rtx_trace_secondary(rtx_scene* s, const rtx_photons* photons, const rtx_shade_point* at, const float* dirs, size_t n, int recursive,
                        int child, uint64_t seed, float* color_dist) {
    if (!s) return rtx_fail(RTX_ERR_INVALID, "scene is NULL");
    if (n == 0) return RTX_OK;
    if (!at || !dirs || !color_dist) return rtx_fail(RTX_ERR_INVALID, "NULL argument");
    if (child < 1 || child > 3) return rtx_fail(RTX_ERR_INVALID, "child must be 1 (diffuse), 2 (reflection) or 3 (refraction)");
    if (at->bounce < 0 || at->diffuse_depth < 0 || at->diffuse_depth > 255) return rtx_fail(RTX_ERR_INVALID, "bad recursion depth");
    if (n > 0x7FFFFFFFull) return rtx_fail(RTX_ERR_TOO_LARGE, "more than 2^31 rays in one call");
    DevPhotonMap pmGlobal, pmCaustic;
    std::memset(&pmGlobal, 0, sizeof(pmGlobal));
    std::memset(&pmCaustic, 0, sizeof(pmCaustic));
    if (photons) {
        photon_device_map(photons, 0, pmGlobal);
        photon_device_map(photons, 1, pmCaustic);
    }
    int rc;
    if ((rc = s->aux.ensure(n * 3 * sizeof(float))) != RTX_OK) return rc;
    if ((rc = s->hits.ensure(n * sizeof(float4))) != RTX_OK) return rc;
    RTX_HIP(hipMemcpy(s->aux.p, dirs, n * 3 * sizeof(float), hipMemcpyHostToDevice));
    launch_legacy_secondary(rtx_traversal_scene(s), s->cfg, pmGlobal, pmCaustic, *at, (const float*)s->aux.p, (uint32_t)n, recursive, child, seed,
                            (float4*)s->hits.p);
    RTX_HIP(hipGetLastError());
    RTX_HIP(hipMemcpy(color_dist, s->hits.p, n * sizeof(float4), hipMemcpyDeviceToHost));
    return RTX_OK;
}

int rtx_film_clear(const rtx_film* film, int width, int height, void* stream) {
    if (!film || !film->accum || !film->count || !film->z || width <= 0 || height <= 0) return rtx_fail(RTX_ERR_INVALID, "bad film");
    if (rtx_device_count() <= 0) return rtx_fail(RTX_ERR_NO_DEVICE, "no HIP device present");
    launch_film_clear(*film, (uint32_t)((size_t)width * height), (hipStream_t)stream);
    RTX_HIP(hipGetLastError());
    return RTX_OK;
}

int rtx_render_accumulate(rtx_scene* s, const rtx_camera* cam, const rtx_render_params* rp, rtx_photons* photons,
                          const rtx_film* film, rtx_stats* stats) {
    return render_impl(s, cam, rp, photons, film, stats, nullptr);
}

int rtx_film_resolve(const rtx_film* film, const rtx_camera* cam, float* linear_rgb, unsigned char* rgb8, float* z, int* samples,
                     float* albedo, float* normal) {
    if (!film || !cam || !film->accum) return rtx_fail(RTX_ERR_INVALID, "bad film");
    if (rtx_device_count() <= 0) return rtx_fail(RTX_ERR_NO_DEVICE, "no HIP device present");
    const size_t n = (size_t)cam->width * cam->height;
    float* dLin = nullptr;
    unsigned char* d8 = nullptr;
    RTX_HIP(hipMalloc((void**)&dLin, n * 3 * sizeof(float)));
    if (hipMalloc((void**)&d8, n * 3) != hipSuccess) {
        (void)hipFree(dLin);
        return rtx_fail(RTX_ERR_HIP, "hipMalloc failed");
    }
    launch_film_resolve(*film, (uint32_t)n, cam->srgb, dLin, d8, nullptr);
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess && linear_rgb) e = hipMemcpy(linear_rgb, dLin, n * 3 * sizeof(float), hipMemcpyDeviceToHost);
    if (e == hipSuccess && rgb8) e = hipMemcpy(rgb8, d8, n * 3, hipMemcpyDeviceToHost);
    if (e == hipSuccess && z) e = hipMemcpy(z, film->z, n * sizeof(float), hipMemcpyDeviceToHost);
    if (e == hipSuccess && samples) e = hipMemcpy(samples, film->count, n * sizeof(int), hipMemcpyDeviceToHost);
    std::vector<float> tmp;
    for (int k = 0; k < 2 && e == hipSuccess; ++k) {
        float* dst = k ? normal : albedo;
        const float* src = k ? film->normal : film->albedo;
        if (!dst) continue;
        if (!src) {
            std::memset(dst, 0, n * 3 * sizeof(float));
            continue;
        }
        tmp.resize(n * 4);
        e = hipMemcpy(tmp.data(), src, n * 4 * sizeof(float), hipMemcpyDeviceToHost);
        for (size_t i = 0; i < n; ++i) {
            dst[3 * i] = tmp[4 * i];
            dst[3 * i + 1] = tmp[4 * i + 1];
            dst[3 * i + 2] = tmp[4 * i + 2];
        }
    }
    (void)hipFree(dLin);
    (void)hipFree(d8);
    if (e != hipSuccess) return rtx_fail(RTX_ERR_HIP, std::string("film resolve: ") + hipGetErrorString(e));
    return RTX_OK;
}

static int alloc_film(size_t n, rtx_film& f, std::vector<void*>& owned) {
    void* p[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    const size_t sz[5] = {n * 16, n * 4, n * 4, n * 16, n * 16};
    for (int k = 0; k < 5; ++k) {
        if (hipMalloc(&p[k], sz[k]) != hipSuccess) {
            for (void* q : owned) (void)hipFree(q);
            owned.clear();
            return rtx_fail(RTX_ERR_HIP, "film allocation failed");
        }
        owned.push_back(p[k]);
    }
    f.accum = (float*)p[0];
    f.count = (int*)p[1];
    f.z = (float*)p[2];
    f.albedo = (float*)p[3];
    f.normal = (float*)p[4];
    return RTX_OK;
}

int rtx_median_denoise(const float* in, int width, int height, int srgb, float* linear_out, unsigned char* rgb8_out) {
    if (!in || width <= 0 || height <= 0) return rtx_fail(RTX_ERR_INVALID, "bad argument");
    if (rtx_device_count() <= 0) return rtx_fail(RTX_ERR_NO_DEVICE, "no HIP device present");
    const size_t n = (size_t)width * height;
    float *dIn = nullptr, *dLin = nullptr;
    unsigned char* d8 = nullptr;
    hipError_t e = hipMalloc((void**)&dIn, n * 12);
    if (e == hipSuccess && linear_out) e = hipMalloc((void**)&dLin, n * 12);
    if (e == hipSuccess && rgb8_out) e = hipMalloc((void**)&d8, n * 3);
    if (e == hipSuccess) e = hipMemcpy(dIn, in, n * 12, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        launch_median3(dIn, width, height, srgb, dLin, d8, nullptr);
        e = hipDeviceSynchronize();
    }
    if (e == hipSuccess && linear_out) e = hipMemcpy(linear_out, dLin, n * 12, hipMemcpyDeviceToHost);
    if (e == hipSuccess && rgb8_out) e = hipMemcpy(rgb8_out, d8, n * 3, hipMemcpyDeviceToHost);
    if (dIn) (void)hipFree(dIn);
    if (dLin) (void)hipFree(dLin);
    if (d8) (void)hipFree(d8);
    if (e != hipSuccess) return rtx_fail(RTX_ERR_HIP, std::string("median denoise: ") + hipGetErrorString(e));
    return RTX_OK;
}

int rtx_film_create(int width, int height, rtx_film* film, rtx_pixel_stats* stats) {
    if (!film || !stats || width <= 0 || height <= 0) return rtx_fail(RTX_ERR_INVALID, "bad argument");
    if (rtx_device_count() <= 0) return rtx_fail(RTX_ERR_NO_DEVICE, "no HIP device present");
    const size_t n = (size_t)width * height;
    std::vector<void*> owned;
    int rc = alloc_film(n, *film, owned);
    if (rc != RTX_OK) return rc;
    stats->sum = stats->sum2 = nullptr;
    if (hipMalloc((void**)&stats->sum, n * 16) != hipSuccess || hipMalloc((void**)&stats->sum2, n * 16) != hipSuccess) {
        rtx_film_destroy(film, stats);
        return rtx_fail(RTX_ERR_HIP, "statistics allocation failed");
    }
    hipError_t e = hipMemset(stats->sum, 0, n * 16);
    if (e == hipSuccess) e = hipMemset(stats->sum2, 0, n * 16);
    if (e != hipSuccess) {
        rtx_film_destroy(film, stats);
        return rtx_fail(RTX_ERR_HIP, "statistics clear failed");
    }
    if ((rc = rtx_film_clear(film, width, height, nullptr)) != RTX_OK) rtx_film_destroy(film, stats);
    return rc;
}

void rtx_film_destroy(rtx_film* film, rtx_pixel_stats* stats) {
    if (film) {
        void* p[5] = {film->accum, film->count, film->z, film->albedo, film->normal};
        for (void* q : p)
            if (q) (void)hipFree(q);
        *film = rtx_film{};
    }
    if (stats) {
        if (stats->sum) (void)hipFree(stats->sum);
        if (stats->sum2) (void)hipFree(stats->sum2);
        stats->sum = stats->sum2 = nullptr;
    }
}

int rtx_render(rtx_scene* s, const rtx_camera* cam, const rtx_render_params* rp, rtx_photons* photons, float* linear_rgb,
               unsigned char* rgb8, float* z, int* samples, rtx_stats* stats) {
    if (!s || !cam || !rp) return rtx_fail(RTX_ERR_INVALID, "NULL argument");
    if (cam->width <= 0 || cam->height <= 0) return rtx_fail(RTX_ERR_INVALID, "invalid camera");
    const size_t n = (size_t)cam->width * cam->height;
    rtx_film f;
    std::vector<void*> owned;
    int rc = alloc_film(n, f, owned);
    if (rc != RTX_OK) return rc;
    rc = rtx_film_clear(&f, cam->width, cam->height, nullptr);
    if (rc == RTX_OK) rc = render_impl(s, cam, rp, photons, &f, stats, nullptr);
    if (rc == RTX_OK) rc = rtx_film_resolve(&f, cam, linear_rgb, rgb8, z, samples, nullptr, nullptr);
    for (void* q : owned) (void)hipFree(q);
    return rc;
}

int rtx_render_samples(rtx_scene* s, const rtx_camera* cam, const rtx_render_params* rp, rtx_photons* photons, float* out,
                       rtx_stats* stats) {
    if (!s || !cam || !rp || !out) return rtx_fail(RTX_ERR_INVALID, "NULL argument");
    if (rp->spp_max != rp->spp_min) return rtx_fail(RTX_ERR_INVALID, "rtx_render_samples needs a fixed sample count");
    if (rp->spp_min < 1) return rtx_fail(RTX_ERR_INVALID, "need 1 <= spp_min <= spp_max");
    if (cam->width <= 0 || cam->height <= 0) return rtx_fail(RTX_ERR_INVALID, "invalid camera");
    const size_t n = (size_t)cam->width * cam->height;
    rtx_film f;
    std::vector<void*> owned;
    int rc = alloc_film(n, f, owned);
    if (rc != RTX_OK) return rc;
    float4* dS = nullptr;
    const size_t bytes = n * (size_t)rp->spp_min * sizeof(float4);
    if (hipMalloc((void**)&dS, bytes) != hipSuccess) {
        for (void* q : owned) (void)hipFree(q);
        return rtx_fail(RTX_ERR_HIP, "sample buffer allocation failed");
    }
    (void)hipMemset(dS, 0, bytes);
    rc = rtx_film_clear(&f, cam->width, cam->height, nullptr);
    if (rc == RTX_OK) rc = render_impl(s, cam, rp, photons, &f, stats, dS);
    if (rc == RTX_OK && hipMemcpy(out, dS, bytes, hipMemcpyDeviceToHost) != hipSuccess) rc = rtx_fail(RTX_ERR_HIP, "copy failed");
    (void)hipFree(dS);
    for (void* q : owned) (void)hipFree(q);
    return rc;
}

}  // extern "C"
