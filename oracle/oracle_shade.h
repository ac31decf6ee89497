// TEST INFRASTRUCTURE ONLY (see oracle_math.h). Light transport of the CPU restatement.
//
// Restates (paths inside the reference repository):
//   PointLight::{RandomPhoton,GenerateSample,GetSampleInfo}   RayTracer/Shading.cpp:88-209
//   BuildPhongContext / SamplePhongBlinn / EvaluatePhongBlinn RayTracer/Shading.cpp:227-713
//   FresnelDielectric, BuildONB, Reflect, TransmissionScale   RayTracer/Shading.cpp:31-61
//   ApplyBumpMapping                                          RayTracer/BumpMapping.h:9-70
//   VisibleSimple / Visible / ApplyAbsorption / DirToScreenUV RayTracer/RayCaster.cpp:178-330
//   TracePath                                                 RayTracer/RayCaster.cpp:339-574
//   splitmix64 / PixelShifts / filters / ToColor24            RayTracer/RayCaster.cpp:33-92,643-690
//   RenderBucket (sampling, splat, adaptive rounds)           RayTracer/RayCaster.cpp:697-1023
//   mergeToImage                                              RayTracer/RayCaster.cpp:1182-1212
//   PhotonMapBuilder::{AdjustPhotonGatherSettings,EmitPhotons,TracePhoton,BuildPhotonMap}
//                                                             RayTracer/PhotonMapBuilder.cpp:20-268
//   EstimateCaustics                                          RayTracer/RayCaster.cpp:143-153
#pragma once
#include "oracle_scene.h"

namespace orc {

enum Lobe { LOBE_NONE = 0, LOBE_DIFFUSE = 1, LOBE_SPECULAR = 2, LOBE_TRANSMISSION = 4, LOBE_ALL = 7 };

// DirSampler::Info
struct DirInfo {
    V3 mult;
    float prob, dist;
    int lobe;
    void set_void() {
        mult = V3(0, 0, 0);
        prob = 0;
        dist = 0;
        lobe = LOBE_NONE;
    }
};

// SamplerInfo as used by the path loop (SetHit/SetPixel/SetPixelSample/SetBounce) with the counter RNG.
struct SInfo {
    V3 P, V, N, GN, uvw;
    bool front = true;
    int mtlID = 0;
    uint64_t seed = 0;
    uint32_t ka = 0, kb = 0, stream = 0;
    float rnd(int dim) const { return rnd01(seed, ka, kb, stream, (uint32_t)dim); }
    void set_hit(const Ray& r, const Hit& h) {
        P = h.p;
        V = -normalized(r.dir);
        N = h.N;
        GN = h.GN;
        uvw = h.uvw;
        front = h.front;
        mtlID = h.mtlID;
    }
};

struct Camera {
    V3 pos, dir, up;
    float fov, focaldist, dof;
    int W, H;
    bool srgb;
};
Camera camera_from(const rtx_camera* c);

// sampler
float halton(int index, int base);
void pixel_shifts(int x, int y, float& sx, float& sy);
Ray generate_ray(const Camera& cam, float samplePosX, float samplePosY, int sampleIndex);
Ray generate_camera_ray(const Camera& cam, int x, int y);  // CameraAdapter.h:7-27
float filter1d(int type, float radiusOverride, float x);
float filter_radius(int type, float radiusOverride);
void to_color24(V3 c, bool srgb, unsigned char out[3]);

// lights
bool light_generate_sample(const Light& L, const SInfo& s, int lightIdx, V3& dir, DirInfo& si);
void light_get_sample_info(const Light& L, V3 P, V3 dir, DirInfo& si);
void light_random_photon(const Light& L, const float u[4], Ray& r, V3& c);
V3 light_radiance(const Light& L);

// materials
bool mtl_generate_sample(const Scene& sc, const Material& m, const SInfo& s, V3& dir, DirInfo& si);
void mtl_get_sample_info(const Scene& sc, const Material& m, const SInfo& s, V3 dir, DirInfo& si);
void apply_bump(const Scene& sc, const Material* m, Hit& h);

float visible(const Scene& sc, V3 origin, V3 normal, V3 dir, float maxDist, bool useTransmittance, uint64_t* hops);

struct PhotonMap {
    std::vector<rtx_photon> ph;
    // uniform grid for neighbour queries
    float cell = 0;
    int imin[3] = {0, 0, 0};
    int gx = 0, gy = 0, gz = 0;
    std::vector<int> cellStart, order;
    void scale(float s);
    void prepare(float radius);
    V3 estimate(int maxPhotons, float radius, V3 pos, V3 normal) const;
};

struct PhotonMapper {
    PhotonMap global, caustic;
    float photonRadius = 0.2f, causticRadius = 0.05f;
    int searchCount = 100;
    uint64_t rays = 0;
};
PhotonMapper* photon_build(const Scene& sc, const rtx_photon_params* p);

struct PathResult {
    V3 radiance;
    float primaryDist;
    V3 albedo, normal;
    bool hit;
};
struct PathCounters {
    uint64_t primary = 0, bounce = 0, shadow = 0;
};
PathResult trace_path(const Scene& sc, const Camera& cam, Ray ray, int px, int py, int sampleIdx,
                      const PhotonMapper* mapper, int maxBounces, uint64_t seed, PathCounters* pc);

V3 miss_radiance(const Scene& sc, const Camera& cam, V3 nd, bool cameraRay);
V3 mtl_diffuse_albedo(const Scene& sc, const Material& m, const SInfo& s);
bool mtl_is_photon_surface(const Material& m);
// legacy shading mode (oracle_legacy.cpp): TraceLegacy, RayCaster.cpp:579-631
PathResult trace_legacy(const Scene& sc, const Camera& cam, Ray ray, int px, int py, int sampleIdx, const PhotonMapper* mapper,
                        uint64_t seed, PathCounters* pc);
V3 legacy_secondary(const Scene& sc, const PhotonMapper* mapper, const rtx_shade_point& at, V3 dir, bool recursive, int child,
                    uint64_t seed, float& dist);
PathResult trace_radiance(const Scene& sc, const Camera& cam, Ray ray, int px, int py, int sampleIdx, const PhotonMapper* mapper,
                          const rtx_render_params& rp, PathCounters* pc);

struct Film {
    int W = 0, H = 0;
    std::vector<float> accum;  // 4 per pixel
    std::vector<int> count;
    std::vector<float> z;
    std::vector<float> albedo, normal;  // 4 per pixel, w = written
    void init(int w, int h);
};
void render(const Scene& sc, const Camera& cam, const rtx_render_params& rp, const PhotonMapper* mapper, Film& film,
            int nthreads, int subset_stride, PathCounters* pc);

}  // namespace orc
