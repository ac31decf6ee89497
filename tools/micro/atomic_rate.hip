// Micro-benchmark: cost of one returning atomicAdd per wave on a single address vs K addresses (queue appends).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k_append(uint32_t* counters, int K, int stride, uint32_t* out, int reps) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint32_t acc = 0;
    for (int r = 0; r < reps; ++r) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&counters[(size_t)((wave + r) % K) * stride], 64u);
        base = __shfl(base, 0);
        acc += base;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
int main() {
    uint32_t *c, *o;
    hipMalloc(&c, 1 << 20);
    hipMalloc(&o, 64);
    hipMemset(c, 0, 1 << 20);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int nWaves = 4 << 20;   // 4M waves
    for (int K : {1, 2, 8, 64, 1024}) {
        for (int reps : {1, 3}) {
            k_append<<<nWaves / 4, 256>>>(c, K, 32, o, reps);
            hipEventRecord(e0);
            k_append<<<nWaves / 4, 256>>>(c, K, 32, o, reps);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("K=%4d reps=%d: %8.3f ms  -> %6.2f ns per atomic\n", K, reps, ms, ms * 1e6 / ((double)nWaves * reps));
        }
    }
    return 0;
}
