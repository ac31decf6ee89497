// Cell of a photon in the gather grid (device code; shared by the counting kernel and the ordering pass)
#pragma once
#include "photon.h"
__device__ __forceinline__ uint32_t cell_of(const PhotonSet& s, float4 p) {
    const int cx = min(s.gx - 1, max(0, (int)((p.x - s.gmin[0]) / s.cell)));
    const int cy = min(s.gy - 1, max(0, (int)((p.y - s.gmin[1]) / s.cell)));
    const int cz = min(s.gz - 1, max(0, (int)((p.z - s.gmin[2]) / s.cell)));
    return (uint32_t)(((size_t)cz * s.gy + cy) * s.gx + cx);
}
