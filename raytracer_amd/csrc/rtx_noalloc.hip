// rtx_refine_list_noalloc (include/rtx.h): the confidence-interval test of rtx_refine_list for callers that cannot
// allocate device memory (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc). It has
// no counterpart in the reference. The scratch of the ordered compaction - one count per block of 1024 pixels, turned
// into exclusive offsets - lives in a file-scope __device__ array instead of a per-call hipMalloc; outputs go to
// device buffers the caller owns; everything is enqueued on the caller's stream and nothing is synchronised.
//
// Each GPU holds its own copy of the array at its own address (hipGetSymbolAddress per device, kept below). Every
// call on a GPU shares that GPU's array: calls that overlap on one GPU (two streams, two threads) race. Nothing in
// it has to start at zero: the count kernel writes every entry the scan and scatter kernels read.
//
// Compiled with IEEE division / square root like kernels_post.hip (adaptive.h).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
#include <string>
#include "adaptive.h"
#include "rtx_internal.h"

#define NA_BLOCK_PIXELS 1024u                       // pixels per compaction block: 256 threads x 4 consecutive pixels
#define NA_MAX_BLOCKS 65536u                        // -> RTX_REFINE_NOALLOC_MAX_PIXELS = 2^26 pixels (8192 x 8192)
#define NA_MAX_DEVICES 64
static_assert((unsigned long long)NA_MAX_BLOCKS * NA_BLOCK_PIXELS == RTX_REFINE_NOALLOC_MAX_PIXELS, "header and kernel limits differ");

// 256 KB + one line per GPU, held until the process ends
static __device__ __align__(256) uint32_t scratch[NA_MAX_BLOCKS + 64];

// refine flags of this thread's 4 consecutive pixels (bit k = pixel base + k)
__device__ __forceinline__ uint32_t na_flags(const float4* __restrict__ sum, const float4* __restrict__ sum2, uint32_t base, uint32_t n,
                                             float maxHalf, int sppMax) {
    uint32_t f = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4u; ++k) {
        const uint32_t pix = base + k;
        if (pix < n && ci_needs_refinement(sum[pix], sum2[pix], maxHalf, sppMax)) f |= 1u << k;
    }
    return f;
}

#if defined(__HIP_DEVICE_COMPILE__) && defined(__AMDGCN_WAVEFRONT_SIZE)
static_assert(__AMDGCN_WAVEFRONT_SIZE == 64, "na_block_scan assumes wave64 (gfx950): lane = tid & 63, four waves per block");
#endif
// exclusive prefix of `v` over the 256 threads of the block (thread order), total in `total`
__device__ __forceinline__ uint32_t na_block_scan(uint32_t v, uint32_t* waveSums, uint32_t& total) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t x = v;
    for (uint32_t d = 1; d < 64u; d <<= 1) {
        const uint32_t y = __shfl_up(x, d);
        if (lane >= d) x += y;
    }
    if (lane == 63u) waveSums[wave] = x;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t w = 0; w < wave; ++w) before += waveSums[w];
    total = waveSums[0] + waveSums[1] + waveSums[2] + waveSums[3];
    return before + x - v;
}

__global__ __launch_bounds__(256) void na_count_kernel(const float4* __restrict__ sum, const float4* __restrict__ sum2, uint32_t n,
                                                       float maxHalf, int sppMax, uint32_t* __restrict__ blockCount) {
    __shared__ uint32_t waveSums[4];
    const uint32_t f = na_flags(sum, sum2, blockIdx.x * NA_BLOCK_PIXELS + threadIdx.x * 4u, n, maxHalf, sppMax);
    uint32_t total;
    (void)na_block_scan((uint32_t)__popc(f), waveSums, total);
    if (threadIdx.x == 0) blockCount[blockIdx.x] = total;
}

// one block: counts -> exclusive offsets in place, total -> *outCount
__global__ __launch_bounds__(256) void na_scan_kernel(uint32_t* __restrict__ blockCount, uint32_t nBlocks, uint32_t* __restrict__ outCount) {
    __shared__ uint32_t waveSums[4];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nBlocks; base += 256u) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < nBlocks ? blockCount[i] : 0u;
        uint32_t total;
        const uint32_t ex = na_block_scan(v, waveSums, total);
        const uint32_t c = carry;
        if (i < nBlocks) blockCount[i] = c + ex;
        __syncthreads();   // everyone has read carry and waveSums
        if (threadIdx.x == 0) carry = c + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) *outCount = carry;
}

__global__ __launch_bounds__(256) void na_scatter_kernel(const float4* __restrict__ sum, const float4* __restrict__ sum2, uint32_t n,
                                                         float maxHalf, int sppMax, const uint32_t* __restrict__ blockOffset,
                                                         int* __restrict__ out, uint32_t capacity) {
    __shared__ uint32_t waveSums[4];
    const uint32_t base = blockIdx.x * NA_BLOCK_PIXELS + threadIdx.x * 4u;
    const uint32_t f = na_flags(sum, sum2, base, n, maxHalf, sppMax);
    uint32_t total;
    uint32_t at = blockOffset[blockIdx.x] + na_block_scan((uint32_t)__popc(f), waveSums, total);
#pragma unroll
    for (uint32_t k = 0; k < 4u; ++k)
        if (f & (1u << k)) {
            if (at < capacity) out[at] = (int)(base + k);
            ++at;
        }
}

// device address of `scratch` on each GPU, looked up once per device with that device current
static std::mutex g_mu;
static uint32_t* g_scratch[NA_MAX_DEVICES];

static int scratch_for_current_device(uint32_t** out) {
    int dev = -1;
    RTX_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= NA_MAX_DEVICES) return rtx_fail(RTX_ERR_INVALID, "device index beyond the scratch table");
    std::lock_guard<std::mutex> lock(g_mu);
    if (!g_scratch[dev]) {
        uint32_t* p = nullptr;
        RTX_HIP(hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch)));
        if (!p) return rtx_fail(RTX_ERR_HIP, "hipGetSymbolAddress returned NULL for the scratch array");
        g_scratch[dev] = p;
    }
    *out = g_scratch[dev];
    return RTX_OK;
}

extern "C" int rtx_refine_list_noalloc(const rtx_pixel_stats* st, int width, int height, float ci_halfwidth, int spp_max,
                                       int* out_pixels_dev, size_t capacity, uint32_t* out_count_dev, void* stream) {
    if (!st || !st->sum || !st->sum2 || width <= 0 || height <= 0 || !out_count_dev || (capacity && !out_pixels_dev))
        return rtx_fail(RTX_ERR_INVALID, "bad argument");
    const unsigned long long n = (unsigned long long)width * (unsigned long long)height;
    if (n > RTX_REFINE_NOALLOC_MAX_PIXELS)
        return rtx_fail(RTX_ERR_TOO_LARGE, "rtx_refine_list_noalloc: width*height = " + std::to_string(n) +
                                               " exceeds RTX_REFINE_NOALLOC_MAX_PIXELS (" +
                                               std::to_string((unsigned long long)RTX_REFINE_NOALLOC_MAX_PIXELS) + ")");
    if (rtx_device_count() <= 0) return rtx_fail(RTX_ERR_NO_DEVICE, "no HIP device present");
    uint32_t* ptr = nullptr;
    int rc = scratch_for_current_device(&ptr);
    if (rc != RTX_OK) return rc;
    const uint32_t nBlocks = (uint32_t)((n + NA_BLOCK_PIXELS - 1u) / NA_BLOCK_PIXELS);
    const uint32_t cap = (uint32_t)(capacity > 0xFFFFFFFFull ? 0xFFFFFFFFull : capacity);
    hipStream_t s = (hipStream_t)stream;
    const float4* sum = (const float4*)st->sum;
    const float4* sum2 = (const float4*)st->sum2;
    hipLaunchKernelGGL(na_count_kernel, dim3(nBlocks), dim3(256), 0, s, sum, sum2, (uint32_t)n, ci_halfwidth, spp_max, ptr);
    hipLaunchKernelGGL(na_scan_kernel, dim3(1), dim3(256), 0, s, ptr, nBlocks, out_count_dev);
    hipLaunchKernelGGL(na_scatter_kernel, dim3(nBlocks), dim3(256), 0, s, sum, sum2, (uint32_t)n, ci_halfwidth, spp_max,
                       (const uint32_t*)ptr, out_pixels_dev, cap);
    RTX_HIP(hipGetLastError());
    return RTX_OK;
}

extern "C" int rtx_device_synchronize(void) {
    if (rtx_device_count() <= 0) return rtx_fail(RTX_ERR_NO_DEVICE, "no HIP device present");
    RTX_HIP(hipDeviceSynchronize());
    return RTX_OK;
}
extern "C" int rtx_stream_create(void** out_stream) {
    if (!out_stream) return rtx_fail(RTX_ERR_INVALID, "out_stream is NULL");
    *out_stream = nullptr;
    if (rtx_device_count() <= 0) return rtx_fail(RTX_ERR_NO_DEVICE, "no HIP device present");
    hipStream_t s = nullptr;
    RTX_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *out_stream = (void*)s;
    return RTX_OK;
}
extern "C" int rtx_stream_synchronize(void* stream) {
    if (rtx_device_count() <= 0) return rtx_fail(RTX_ERR_NO_DEVICE, "no HIP device present");
    RTX_HIP(hipStreamSynchronize((hipStream_t)stream));
    return RTX_OK;
}
extern "C" int rtx_stream_destroy(void* stream) {
    if (!stream) return RTX_OK;
    RTX_HIP(hipStreamDestroy((hipStream_t)stream));
    return RTX_OK;
}
