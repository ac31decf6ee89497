// Radix-sort passes of the photon map build (rocPRIM): storage order of a map and its capacity rule. (The ray and shade
// regrouping experiments of round 2 that shared this file - measured negatives, DESIGN.md section 3 - left the tree in
// round 3; `git log -- raytracer_amd/csrc/kernels_sort.hip` has them.)
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include "kernels.h"
#include "pathstate.h"
#include "photon_cell.h"

// ---- photon map order ------------------------------------------------------------------------------------------
// The photons of a map are stored cell by cell and, inside a cell, by their (photon, bounce) key: the emit kernel
// appends records in whatever order its lanes finish, and the gather sums in storage order, so without the second key
// two builds of one map - or the maps of one GPU and of eight - would differ in the last bits of every estimate.
__global__ void photon_key_kernel(PhotonSet s, unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= s.n) return;
    const float4 p = s.raw[i].pos;
    keys[i] = ((unsigned long long)cell_of(s, p) << 32) | (unsigned long long)__float_as_uint(p.w);
    vals[i] = i;
}
// Power is scaled here (PhotonMapBuilder.cpp:38-50)
__global__ void photon_place_kernel(PhotonSet s, const uint32_t* __restrict__ order, float scale) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= s.n) return;
    const DevPhotonRec r = s.raw[order[k]];
    s.pos[k] = r.pos;
    s.dir[k] = r.dir;
    s.power[k] = make_float4(r.power.x * scale, r.power.y * scale, r.power.z * scale, 0.0f);
}
int launch_photon_order(const PhotonSet& s, float scale, hipStream_t st) {
    if (!s.n) return 0;
    unsigned long long *keysIn = nullptr, *keysOut = nullptr;
    uint32_t *valsIn = nullptr, *valsOut = nullptr;
    void* temp = nullptr;
    size_t bytes = 0;
    unsigned cellBits = 1;
    while (cellBits < 32 && (1ull << cellBits) < (unsigned long long)s.cells) ++cellBits;
    int rc = 0;
    auto fail = [&](hipError_t e) { if (e != hipSuccess && !rc) rc = (int)e; return e != hipSuccess; };
    if (fail(hipMalloc((void**)&keysIn, (size_t)s.n * 8)) || fail(hipMalloc((void**)&keysOut, (size_t)s.n * 8)) ||
        fail(hipMalloc((void**)&valsIn, (size_t)s.n * 4)) || fail(hipMalloc((void**)&valsOut, (size_t)s.n * 4)) ||
        fail(rocprim::radix_sort_pairs(nullptr, bytes, (const unsigned long long*)keysIn, keysOut, (const uint32_t*)valsIn, valsOut, s.n, 0u,
                                       32u + cellBits, st)) ||
        fail(hipMalloc(&temp, bytes ? bytes : 1))) {
        // fall through to the releases below
    } else {
        hipLaunchKernelGGL(photon_key_kernel, dim3((s.n + 255) / 256), dim3(256), 0, st, s, keysIn, valsIn);
        if (!fail(rocprim::radix_sort_pairs(temp, bytes, (const unsigned long long*)keysIn, keysOut, (const uint32_t*)valsIn, valsOut, s.n, 0u,
                                            32u + cellBits, st))) {
            hipLaunchKernelGGL(photon_place_kernel, dim3((s.n + 255) / 256), dim3(256), 0, st, s, (const uint32_t*)valsOut, scale);
            fail(hipStreamSynchronize(st));
        }
    }
    (void)hipFree(keysIn);
    (void)hipFree(keysOut);
    (void)hipFree(valsIn);
    (void)hipFree(valsOut);
    (void)hipFree(temp);
    return rc;
}

// ---- photon map capacity -----------------------------------------------------------------------------------------
// PhotonMap::Resize(numPhotons) / Resize(numPhotons / 10) (PhotonMapBuilder.cpp:27-28): a full map takes no more
// photons. Which ones the reference drops depends on its thread timing; here the records with the smallest
// (photon, bounce) keys stay - what emission in photon order would have stored - so the result does not depend on the
// order in which the emit kernel's lanes finished, nor on how many GPUs emitted.
__global__ void photon_idkey_kernel(const DevPhotonRec* __restrict__ raw, uint32_t n, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    keys[i] = __float_as_uint(raw[i].pos.w);
    vals[i] = i;
}
__global__ void photon_pick_kernel(const DevPhotonRec* __restrict__ raw, const uint32_t* __restrict__ order, uint32_t keep,
                                   DevPhotonRec* __restrict__ out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= keep) return;
    out[k] = raw[order[k]];
}
// s.raw (s.n records) -> a new buffer of the `keep` records with the smallest keys; returns 0 or a hipError_t
int launch_photon_keep_smallest(PhotonSet& s, uint32_t keep, hipStream_t st) {
    if (keep >= s.n) return 0;
    if (keep == 0) {
        s.n = 0;
        return 0;
    }
    uint32_t *keysIn = nullptr, *keysOut = nullptr, *valsIn = nullptr, *valsOut = nullptr;
    DevPhotonRec* out = nullptr;
    void* temp = nullptr;
    size_t bytes = 0;
    int rc = 0;
    auto fail = [&](hipError_t e) { if (e != hipSuccess && !rc) rc = (int)e; return e != hipSuccess; };
    if (fail(hipMalloc((void**)&keysIn, (size_t)s.n * 4)) || fail(hipMalloc((void**)&keysOut, (size_t)s.n * 4)) ||
        fail(hipMalloc((void**)&valsIn, (size_t)s.n * 4)) || fail(hipMalloc((void**)&valsOut, (size_t)s.n * 4)) ||
        fail(hipMalloc((void**)&out, (size_t)keep * sizeof(DevPhotonRec))) ||
        fail(rocprim::radix_sort_pairs(nullptr, bytes, (const uint32_t*)keysIn, keysOut, (const uint32_t*)valsIn, valsOut, s.n, 0u, 32u, st)) ||
        fail(hipMalloc(&temp, bytes ? bytes : 1))) {
        (void)hipFree(out);
        out = nullptr;
    } else {
        hipLaunchKernelGGL(photon_idkey_kernel, dim3((s.n + 255) / 256), dim3(256), 0, st, (const DevPhotonRec*)s.raw, s.n, keysIn, valsIn);
        if (!fail(rocprim::radix_sort_pairs(temp, bytes, (const uint32_t*)keysIn, keysOut, (const uint32_t*)valsIn, valsOut, s.n, 0u, 32u, st))) {
            hipLaunchKernelGGL(photon_pick_kernel, dim3((keep + 255) / 256), dim3(256), 0, st, (const DevPhotonRec*)s.raw, (const uint32_t*)valsOut,
                               keep, out);
            fail(hipStreamSynchronize(st));
        }
        if (rc) {
            (void)hipFree(out);
            out = nullptr;
        }
    }
    (void)hipFree(keysIn);
    (void)hipFree(keysOut);
    (void)hipFree(valsIn);
    (void)hipFree(valsOut);
    (void)hipFree(temp);
    if (out) {
        (void)hipFree(s.raw);
        s.raw = out;
        s.n = keep;
        s.capacity = keep;
    }
    return rc;
}
