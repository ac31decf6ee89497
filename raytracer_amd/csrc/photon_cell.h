// Cell of a photon in the gather grid (device code; shared by the counting kernel and the ordering pass).
// The grid is a window [imin, imin + g) of the lattice floor(p / cell) anchored at the origin (rtx_photon.cpp finalize_set).
#pragma once
#include "photon.h"
__device__ __forceinline__ int lattice_clamped(float x, float cell, int imin, int g) {
    return (int)fminf(fmaxf(floorf(x / cell) - (float)imin, 0.0f), (float)(g - 1));
}
__device__ __forceinline__ uint32_t cell_of(const PhotonSet& s, float4 p) {
    const int cx = lattice_clamped(p.x, s.cell, s.imin[0], s.gx);
    const int cy = lattice_clamped(p.y, s.cell, s.imin[1], s.gy);
    const int cz = lattice_clamped(p.z, s.cell, s.imin[2], s.gz);
    return (uint32_t)(((size_t)cz * s.gy + cy) * s.gx + cx);
}
