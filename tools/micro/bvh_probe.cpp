// Host-only probe of the tree builder and the traversal-order model (bvh_build.cpp), built once per tree layout:
//   g++ -O2 -std=c++17 -fPIC -shared [-DRTX_BVH8=1] -I raytracer_amd/csrc tools/micro/bvh_probe.cpp raytracer_amd/csrc/bvh_build.cpp -o /tmp/libbvhprobeN.so -pthread
// tools/bvh_walk_probe.py loads both and compares steps per ray on the bench scene. Measurement aid, not a product path.
#include "bvh_build.h"
#include <chrono>
#include <thread>
extern "C" int bvh_probe(const rtx_scene_desc* desc, const float* rays8, size_t n, float* t_out, int* prim_out, uint64_t counts[6]) {
    HostScene hs;
    std::string err;
    if (!build_host_scene(desc, hs, err)) return 1;
    err = validate_host_scene(hs);
    if (!err.empty()) {
        fprintf(stderr, "validate: %s\n", err.c_str());
        return 2;
    }
    counts[2] = hs.nodes.size();
    counts[3] = (uint64_t)hs.depth;
    counts[4] = sizeof(BvhNode);
    const int NT = 8;
    uint64_t part[NT][2] = {};
    std::vector<std::thread> th;
    for (int k = 0; k < NT; ++k)
        th.emplace_back([&, k]() {
            for (size_t i = k; i < n; i += NT) host_walk_closest(hs, rays8 + 8 * i, t_out + i, prim_out + i, part[k]);
        });
    for (auto& t : th) t.join();
    counts[0] = counts[1] = 0;
    for (int k = 0; k < NT; ++k) {
        counts[0] += part[k][0];
        counts[1] += part[k][1];
    }
    return 0;
}
