// Segmented work queues between the wavefront kernels.
//
// A queue append needs one returning atomic per wave, and atomics on ONE address are served at about 11 ns each on
// MI355X (tools/micro/atomic_rate.hip): with ~10^7 waves per frame a single queue counter alone would cost more than the
// shading it feeds. A SegQ therefore has QSEG independent segments, each with its own counter in its own 128-byte
// line. A producer block appends to the segment its block index selects; the segment's region of the item array
// starts where the first block of that segment would write if nothing were dropped, so regions never overlap and
// every segment is dense. segq_scan_kernel turns the counters into an exclusive prefix (and the total, for the host),
// and consumers either map a flat index through the prefix (one thread per item) or drain the segments through
// per-segment cursors (persistent kernels).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>

#define QSEG 64u          // segments per queue (= lanes of a wave: one lane scans / inspects one segment)
#define QSTRIDE 32u       // uint32 stride between per-segment counters: one 128-byte line each
#define SEGQ_COUNT 0u                       // [QSEG * QSTRIDE] appended entries per segment (producers, atomics)
#define SEGQ_CURSOR (QSEG * QSTRIDE)        // [QSEG * QSTRIDE] consumed entries per segment (persistent consumers)
#define SEGQ_PREFIX (2u * QSEG * QSTRIDE)   // [QSEG + 1] exclusive prefix of the counts; [QSEG] = total
#define SEGQ_START (SEGQ_PREFIX + QSEG + 16u)   // [QSEG] first item index of each segment's region
#define SEGQ_WORDS (SEGQ_START + QSEG + 16u)

struct SegQ {
    uint32_t* ctl;        // SEGQ_WORDS control words (layout above)
    uint32_t run = 0u;    // 0: contiguous segments; K: producer blocks go round the segments in runs of K (segq_segment_of_block)
};

// Segment of producer block b in a grid of G blocks, and the number of blocks in the segments before s (the first block of a
// contiguous segment). run = 0: segment s holds the blocks [s G / 64, (s + 1) G / 64): the queue keeps the producers' order.
// run = K > 0: the blocks go round the segments in runs of K - the blocks in flight at one time, whose indices are neighbours,
// then append through 64 counters instead of one or two (SegQ::run).
__host__ __device__ inline uint32_t segq_segment_of_block(uint32_t b, uint32_t G, uint32_t run = 0u) {
    if (run) return (b / run) % QSEG;
    return (uint32_t)(((unsigned long long)b * QSEG) / G);
}
__host__ __device__ inline uint32_t segq_first_block(uint32_t s, uint32_t G, uint32_t run = 0u) {
    if (run) {
        const uint32_t runs = G / run, rem = G % run, r = runs % QSEG;   // full runs; blocks of the last, partial run (of segment r)
        return run * (s * (runs / QSEG) + (s < r ? s : r)) + (r < s ? rem : 0u);
    }
    return (uint32_t)(((unsigned long long)s * G + QSEG - 1u) / QSEG);
}

#ifdef __HIPCC__
// Producer side: every lane of the wave calls this (pred = lane has an entry); returns the item index to write.
// mult = entries a producer thread can append at most (region size per block = blockDim.x * mult).
// G: the number of producer blocks the queue's regions are laid out for. It is the grid size, unless the grid is an upper bound
// whose tail blocks produce nothing (nee_kernel): all entries would then crowd into the first few segments, and the persistent
// consumers, which start spread over the 64 segments, would queue up behind a handful of cursors.
__device__ __forceinline__ uint32_t segq_append_g(const SegQ& q, bool pred, uint32_t mult, uint32_t G);
__device__ __forceinline__ uint32_t segq_append(const SegQ& q, bool pred, uint32_t mult) { return segq_append_g(q, pred, mult, gridDim.x); }
__device__ __forceinline__ uint32_t segq_append_g(const SegQ& q, bool pred, uint32_t mult, uint32_t G) {
    const uint32_t seg = segq_segment_of_block(blockIdx.x, G, q.run);
    const unsigned long long mask = __ballot(pred);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t total = (uint32_t)__popcll(mask);
    uint32_t base = 0;
    if (total) {
        const int leader = __ffsll((long long)mask) - 1;
        if ((int)lane == leader) base = atomicAdd(&q.ctl[SEGQ_COUNT + seg * QSTRIDE], total);
        base = __shfl(base, leader);
    }
    const uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    return segq_first_block(seg, G, q.run) * blockDim.x * mult + base + prefix;
}

// The same with one atomic per BLOCK: every thread of the block calls it, outside divergent control flow (two barriers). The
// blocks that run at one time append to the same segment or two (their indices are neighbours), so the appends of a kernel queue
// up at one address after all; a block of four waves asks a quarter as often. tmp: SEGQ_BLOCK_WORDS words of LDS.
#define SEGQ_BLOCK_WORDS 20u   // [0] base, [1] block total, [2 + w] entries of wave w (blocks of up to 16 waves)
__device__ __forceinline__ uint32_t segq_append_block(const SegQ& q, bool pred, uint32_t mult, uint32_t G, uint32_t* tmp) {
    const uint32_t seg = segq_segment_of_block(blockIdx.x, G, q.run);
    const unsigned long long mask = __ballot(pred);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nWaves = (blockDim.x + 63u) >> 6;
    if (lane == 0) tmp[2u + wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (uint32_t w = 0; w < nWaves; ++w) total += tmp[2u + w];
        tmp[0] = total ? atomicAdd(&q.ctl[SEGQ_COUNT + seg * QSTRIDE], total) : 0u;
    }
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t w = 0; w < wave; ++w) before += tmp[2u + w];
    const uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    return segq_first_block(seg, G, q.run) * blockDim.x * mult + tmp[0] + before + prefix;
}

// The same for a wave that appends `total` entries at once (wave-uniform, > 0; every active lane calls): returns the item
// index of the first; the caller hands out [first, first + total) among its lanes.
__device__ __forceinline__ uint32_t segq_reserve(const SegQ& q, uint32_t total, uint32_t mult) {
    const uint32_t seg = segq_segment_of_block(blockIdx.x, gridDim.x, q.run);
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long mask = __ballot(true);
    const int leader = __ffsll((long long)mask) - 1;
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(&q.ctl[SEGQ_COUNT + seg * QSTRIDE], total);
    base = __shfl(base, leader);
    return segq_first_block(seg, gridDim.x, q.run) * blockDim.x * mult + base;
}

// Flat consumer: item index of entry i (i < total). tab = LDS copy of prefix[QSEG+1] followed by start[QSEG],
// filled by segq_load_table (all threads of the block call it, then __syncthreads()).
#define SEGQ_TABLE_WORDS (2u * QSEG + 1u)
__device__ __forceinline__ void segq_load_table(const SegQ& q, uint32_t* tab) {
    for (uint32_t k = threadIdx.x; k < SEGQ_TABLE_WORDS; k += blockDim.x)
        tab[k] = k <= QSEG ? q.ctl[SEGQ_PREFIX + k] : q.ctl[SEGQ_START + (k - QSEG - 1u)];
}
__device__ __forceinline__ uint32_t segq_item_index(const uint32_t* tab, uint32_t i) {
    uint32_t s = 0;
#pragma unroll
    for (uint32_t step = QSEG / 2u; step > 0u; step >>= 1)
        if (tab[s + step] <= i) s += step;   // largest s with prefix[s] <= i
    return tab[QSEG + 1u + s] + (i - tab[s]);
}

// First segment of a persistent consumer wave. Waves start in SEGQ_GROUPS groups spread over the queue and walk
// forward from there: enough distinct counters to keep the atomics off one address, few enough active regions that
// the waves running at the same time work on neighbouring rays (which share BVH nodes in L2).
// (re-tuned in round 3, once the producers appended per block - C3 frame, extend / shadow ms: 4 groups 313 / 178, 8: 295 / 161,
// 16: 287.5 / 152.7, 32: 283.7 / 149.4, 64: 285 / 150; then reservations of at most 2048 / 1024 / 512 / 256 entries with 32
// groups: extend 290 / 285 / 281 / 281, shadow 152 / 150.3 / 148.3 / 150.4. The waves' waits at the cursors are what the
// smaller groups and reservations save; 48 groups, which do not divide the segments evenly: 306 / 174.)
#ifndef SEGQ_GROUPS
#define SEGQ_GROUPS 32u
#endif
#ifndef SEGQ_CHUNK_MAX
#define SEGQ_CHUNK_MAX 512ull
#endif
#ifndef SEGQ_CHUNK_DIV
#define SEGQ_CHUNK_DIV 8ull
#endif
__device__ __forceinline__ uint32_t segq_home(uint32_t wave) { return (wave % SEGQ_GROUPS) * (QSEG / SEGQ_GROUPS); }

// Persistent consumer. All fields are wave-uniform. A wave reserves `chunk` consecutive entries of its current
// segment with one atomic and hands them to its lanes over several refills, so most refills cost no atomic round
// trip (during which the wave's busy lanes would stall too).
// Two ways round the queue, an instantiation each (FRONT), chosen per launch by its size (segq_front):
//  - by segment (large launches): as described at segq_home - 32 stretches of the producers' order in work at a time.
//  - as one front (launches of a few hundred thousand to a few million entries, where a wave gets a handful of entries per lane
//    and the launch is as long as its unluckiest wave): chunk c of the queue is chunk c / QSEG of segment c % QSEG, every wave
//    starts with the chunk of its own index - no atomic, no wave without work while another holds two chunks - and takes further
//    ones from ONE cursor. Extend and shadow launches of 0.25 - 5 M rays: 10 - 20 % shorter; larger ones lose 8 % (64 places of
//    the producers' order in work at a time), and both walks compiled into one kernel cost extend_kernel two spilled registers.
struct SegWalk {
    uint32_t seg;              // current segment
    uint32_t cnt, start;       // its entry count and first item index
    uint32_t resBase, resEnd;  // reserved, not yet handed out entries [resBase, resEnd) of the segment
    uint32_t chunk;            // entries reserved per atomic (multiple of 64)
    uint32_t limit, next, nWaves;   // FRONT: chunks of the queue (QSEG x chunks of the longest segment), this wave's next chunk, waves of the grid
};
__device__ __forceinline__ void segq_walk_load(const SegQ& q, SegWalk& w) {
    w.cnt = __builtin_amdgcn_readfirstlane(q.ctl[SEGQ_PREFIX + w.seg + 1u] - q.ctl[SEGQ_PREFIX + w.seg]);
    w.start = __builtin_amdgcn_readfirstlane(q.ctl[SEGQ_START + w.seg]);
    w.resBase = w.resEnd = 0u;
}
template <bool FRONT = false>
__device__ __forceinline__ void segq_walk_init(const SegQ& q, SegWalk& w, uint32_t wave, uint32_t chunk) {
    w.chunk = chunk;
    w.limit = w.next = w.nWaves = 0u;
    if (FRONT) {
        const uint32_t lane = threadIdx.x & 63u;
        uint32_t c = q.ctl[SEGQ_PREFIX + lane + 1u] - q.ctl[SEGQ_PREFIX + lane];
        for (uint32_t d = 32u; d > 0u; d >>= 1) c = max(c, (uint32_t)__shfl_xor((int)c, (int)d));
        w.limit = QSEG * ((__builtin_amdgcn_readfirstlane(c) + chunk - 1u) / chunk);
        w.next = wave;
        w.nWaves = (gridDim.x * blockDim.x) >> 6;
        w.seg = w.cnt = w.start = w.resBase = w.resEnd = 0u;
        return;
    }
    w.seg = segq_home(wave);
    segq_walk_load(q, w);
}
// The wave asks for `want` <= 64 entries. Returns the number granted (0 = the queue is drained: every segment's cursor
// has passed its count) and the item index of the first one in `first`; granted entries are consecutive. Fewer than
// `want` may be granted when a reservation runs out; the caller asks again at its next refill.
template <bool FRONT = false>
__device__ __forceinline__ uint32_t segq_take(const SegQ& q, SegWalk& w, uint32_t want, uint32_t& first) {
    const uint32_t lane = threadIdx.x & 63u;
    for (;;) {
        if (w.resBase < w.resEnd) {
            const uint32_t g = min(want, w.resEnd - w.resBase);
            first = w.start + w.resBase;
            w.resBase += g;
            return g;
        }
        if (FRONT) {
            if (w.next >= w.limit) return 0u;
            const uint32_t c = w.next;
            uint32_t t = 0;   // the chunk after this one: the shared cursor counts the chunks beyond the waves' first ones
            if (lane == 0) t = atomicAdd(&q.ctl[SEGQ_CURSOR], 1u);
            w.next = w.nWaves + __builtin_amdgcn_readfirstlane(t);
            w.seg = c % QSEG;
            const uint32_t off = (c / QSEG) * w.chunk;
            const uint32_t cnt = __builtin_amdgcn_readfirstlane(q.ctl[SEGQ_PREFIX + w.seg + 1u] - q.ctl[SEGQ_PREFIX + w.seg]);
            if (off < cnt) {
                w.start = __builtin_amdgcn_readfirstlane(q.ctl[SEGQ_START + w.seg]);
                w.resBase = off;
                w.resEnd = min(cnt, off + w.chunk);
            }
            continue;
        }
        if (w.cnt) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&q.ctl[SEGQ_CURSOR + w.seg * QSTRIDE], w.chunk);
            base = __builtin_amdgcn_readfirstlane(base);
            if (base < w.cnt) {
                w.resBase = base;
                w.resEnd = min(w.cnt, base + w.chunk);
                continue;
            }
        }
        // this segment is drained: lane l looks at segment l, continue with the next one that still has entries
        const uint32_t c = q.ctl[SEGQ_PREFIX + lane + 1u] - q.ctl[SEGQ_PREFIX + lane];
        const uint32_t cur = __atomic_load_n(&q.ctl[SEGQ_CURSOR + lane * QSTRIDE], __ATOMIC_RELAXED);
        const unsigned long long open = __ballot(cur < c);
        if (open == 0ull) return 0u;
        const unsigned long long above = w.seg < 63u ? (open >> (w.seg + 1u)) << (w.seg + 1u) : 0ull;
        w.seg = (uint32_t)(__ffsll((long long)(above ? above : open)) - 1);
        segq_walk_load(q, w);
    }
}
#endif

// Host-callable wrappers (kernels_trace.hip)
// counts -> prefix/start/total for a queue produced by a grid of G blocks of B threads appending <= mult entries per
// thread; zeroes the counters and cursors; totals[slot] = number of entries.
// blocksFrom (optional): a queue total on the device; the producer grid then counts ceil(*blocksFrom / blockThreads) blocks
// (a producer launched over an upper bound that lays its regions out by the true count: nee_kernel)
// blocksFrom: the producer is nee_kernel, which runs its blocks in a permuted order over a grid rounded up to QSEG rows
// (segq_nee_rows): G = that grid
void launch_segq_scan(const SegQ& q, uint32_t producerBlocks, uint32_t blockThreads, uint32_t mult, uint32_t* totals,
                      uint32_t totalsSlot, hipStream_t stream, const uint32_t* blocksFrom = nullptr);
// nee_kernel's grid for `total` queued pairs in blocks of B threads: QSEG x rows blocks, block j works on the flat chunk
// (j % QSEG) * rows + j / QSEG (kernels_nee.hip)
__host__ __device__ inline uint32_t segq_nee_rows(uint32_t total, uint32_t B) { return ((total + B - 1u) / B + QSEG - 1u) / QSEG; }
// identity queue over [0, n): segment s covers [s*n/QSEG, (s+1)*n/QSEG)
void launch_segq_identity(const SegQ& q, uint32_t n, hipStream_t stream);
// whether a launch over n entries walks its queue as one front (SegWalk); RTX_AMD_FRONT_MIN / _MAX move the limits
inline bool segq_front(uint32_t n) {
    const char* a = std::getenv("RTX_AMD_FRONT_MIN");   // (read per launch: the tests switch them between renders)
    const char* b = std::getenv("RTX_AMD_FRONT_MAX");
    const long long lo = a ? std::atoll(a) : 230000LL, hi = b ? std::atoll(b) : 5500000LL;
    return (long long)n >= lo && (long long)n <= hi;
}
// entries a persistent consumer wave reserves per atomic, for n entries drained by `waves` waves
inline uint32_t segq_chunk(uint32_t n, uint32_t waves) {
    unsigned long long c = (unsigned long long)n / ((unsigned long long)(waves ? waves : 1u) * SEGQ_CHUNK_DIV);
    c = (c / 64ull) * 64ull;
    return (uint32_t)(c < 64ull ? 64ull : (c > SEGQ_CHUNK_MAX ? SEGQ_CHUNK_MAX : c));
}
