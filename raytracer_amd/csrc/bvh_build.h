// Host-side scene flattening: (Node,TriObj) pairs -> one quantised 4-wide BVH + compact triangle streams.
// Replaces TriObj::BuildAcceleration (reference RayTracer/objects.cpp:128-138) and the scene-graph walk of
// IntersectNodeLocal (RayTracer/IntersectNodeAdapter.h:23-97) by a single flat structure (layout.h).
#pragma once
#include <string>
#include <vector>
#include "layout.h"
#include "../../include/rtx.h"

struct HostScene {
    std::vector<BvhNode> nodes;
    std::vector<TriRec> tris;
    std::vector<TriAttr> attrs;
    std::vector<ObjRec> objs;
    std::vector<ObjXform> objXf;
    uint32_t root = RTX_EMPTY_CHILD;
    float bounds[6] = {1e30f, 1e30f, 1e30f, -1e30f, -1e30f, -1e30f};
    int depth = 0;
};

// Returns false and sets err on invalid input (index out of range, non-finite vertex, too many faces).
bool build_host_scene(const rtx_scene_desc* desc, HostScene& out, std::string& err);

// Structural self-check used by the CPU tests: every triangle in exactly one leaf, every child box contains
// its subtree, quantised boxes are conservative. Returns an empty string when consistent.
std::string validate_host_scene(const HostScene& hs);

// Host model of the kernels' traversal order for one ray {o.xyz, tmax, d.xyz, -}: closest hit (t and global triangle id, -1 / -1
// for a miss) among the world-space triangles, counts[0] += nodes visited, counts[1] += triangles tested. Test and
// measurement aid (rtx_debug_host_walk); no render path calls it.
void host_walk_closest(const HostScene& hs, const float* ray8, float* tOut, int* primOut, uint64_t counts[2]);
