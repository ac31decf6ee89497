/*
 * rtx.h — C ABI of the MI355X-native ray-casting / light-transport path.
 *
 * This is the drop-in boundary for the hot path of DHRRAM/RayTracer. The reference has no FFI layer of its
 * own (it is one C++ executable); each entry point below names the in-process C++ interface it replaces
 * (paths relative to the reference repository). The C++ host mirror in raytracer_amd/host/ keeps the
 * reference's class and function names (RayCaster, IntersectNode, IntersectSceneWithLights,
 * CollectSceneNodeObjects, ShadeInfoAdapter, PhotonMapBuilder) and forwards to these functions.
 *
 * Conventions
 *  - Only POD structs and C types cross the boundary. Status 0 = ok, negative = error; the message is
 *    available from rtx_last_error() (thread-local).
 *  - Host pointers in descriptors are caller-owned and copied before the call returns.
 *  - Handles own device memory. One in-flight call per handle; different handles may be used from different
 *    threads.
 *  - There is no CPU fallback: every compute entry point fails with RTX_ERR_NO_DEVICE when no HIP device is
 *    present.
 */
#ifndef RTX_H
#define RTX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- constants (framework conventions used by the reference's hot path) ------------------------------- */
#define RTX_HIT_NONE 0
#define RTX_HIT_FRONT 1            /* IntersectNodeAdapter.h:100 default hitSide                            */
#define RTX_HIT_BACK 2
#define RTX_HIT_FRONT_AND_BACK 3   /* RayCaster.cpp:361 (path loop), RayCaster.h:82                          */
#define RTX_BIGFLOAT 1.0e30f       /* framework BIGFLOAT (stand-in choice)                   */

#define RTX_OK 0
#define RTX_ERR_INVALID (-1)
#define RTX_ERR_NO_DEVICE (-2)
#define RTX_ERR_HIP (-3)
#define RTX_ERR_TOO_LARGE (-4)

#define RTX_MAX_LIGHTS 8

#define RTX_OBJ_MESH 0             /* TriObj,  objects.cpp:128-266 */
#define RTX_OBJ_SPHERE 1           /* Sphere,  objects.cpp:21-86   */
#define RTX_OBJ_PLANE 2            /* Plane,   objects.cpp:89-126  */

#define RTX_MTL_BLINN 0            /* Shading.cpp:715-725 */
#define RTX_MTL_PHONG 1            /* Shading.cpp:727-737 */
#define RTX_MTL_MICROFACET 2       /* MtlMicrofacet, Shading.cpp:741-936 (FBX import): diffuse = baseColor,
                                      specular[0] = roughness, reflection[0] = metallic, refraction = transmittance
                                      (the *_tex slots are their maps: a float map scales by the texel's red
                                      channel), ior, absorption, bump as for the other types; emission is ignored
                                      (RayCaster.cpp:158-161) and the albedo guide is 0.5 (:135-138) */

#define RTX_TEX_RGB8 0             /* Color24 texels (PNG)            */
#define RTX_TEX_RGB32F 1           /* float texels (Radiance HDR)     */

#define RTX_FILTER_TENT 0          /* RayCaster.h:23, RayCaster.cpp:643-690 */
#define RTX_FILTER_GAUSSIAN 1
#define RTX_FILTER_MITCHELL 2
#define RTX_FILTER_LANCZOS 3
#define RTX_MAX_FILTER_RADIUS 16.0f /* largest filter_radius accepted (the reference's supports are 1 .. 2 pixels, :84-92) */

typedef struct rtx_scene rtx_scene;   /* opaque: flattened BVH + triangles + materials + lights in HBM */
typedef struct rtx_photons rtx_photons; /* opaque: photon maps + hash grids in HBM                      */

/* ---- scene description -------------------------------------------------------------------------------- */

/* One (Node, TriObj) pair as produced by CollectSceneNodeObjects (SceneAdapter.h:8-18). Index triplets and
 * vertex arrays follow cy::TriMesh (V/VN/VT, F/FN/FT). vn/fn and vt/ft may be NULL (objects.cpp:181,202). */
typedef struct {
    const float* v;   int nv;      /* xyz                                                    */
    const float* vn;  int nvn;     /* xyz, optional                                          */
    const float* vt;  int nvt;     /* uvw, optional                                          */
    const int* f;                  /* 3*nf vertex indices                                    */
    const int* fn;                 /* 3*nf normal indices, optional (NULL: same as f)        */
    const int* ft;                 /* 3*nf texcoord indices, optional (NULL: same as f)      */
    const int* face_mtl;           /* nf per-face GetMaterialIndex, optional (objects.cpp:236) */
    int nf;
    float xform[12];               /* object->world 3x4 row-major (accumulated Node transforms; identity for
                                      every object of assets/cerealBar.xml:5-51). Must be finite. A mirroring
                                      transform (det < 0) keeps the object's own facing: front/back, GN and
                                      rtx_hit.bu/bv follow the mesh's vertex order as in the reference, which
                                      intersects in object space (IntersectNodeAdapter.h:62,93) */
    int material;                  /* index into rtx_scene_desc.materials; -1 = none         */
    int kind;                      /* RTX_OBJ_MESH (arrays above), or an analytic object in object space under
                                      xform: RTX_OBJ_SPHERE = unit sphere (objects.cpp:21-86), RTX_OBJ_PLANE = the
                                      square [-1,1]^2 at z = 0 (objects.cpp:89-126); arrays are ignored for those */
} rtx_mesh_desc;

/* Texture image + TextureMap transform (texture-space matrix applied to uvw before lookup). */
typedef struct {
    int format;                    /* RTX_TEX_*            */
    int width, height;
    const void* texels;            /* width*height*3 u8 or float, row 0 first */
    float xform[12];               /* uvw' = M * (uvw,1), 3x4 row-major       */
} rtx_texmap_desc;

/* MtlBlinn / MtlPhong parameters (framework MtlBasePhongBlinn; Shading.cpp:227-278).
 * *_tex = index into texmaps or -1. The TexturedColor value multiplies the texture. */
typedef struct {
    int type;                      /* RTX_MTL_*                                     */
    float diffuse[3];    int diffuse_tex;
    float specular[3];   int specular_tex;
    float reflection[3]; int reflection_tex;
    float refraction[3]; int refraction_tex;
    float emission[3];   int emission_tex;
    float glossiness;
    int glossiness_tex;            /* TexturedFloat map of Glossiness() (Shading.cpp:238 sInfo.Eval(mtl.Glossiness()),
                                      Shading_Legacy.cpp:375): value * red channel of the texel; -1 = none */
    float ior;
    float absorption[3];           /* Beer-Lambert coefficients, RayCaster.cpp:281-301 */
    int bump_tex;                  /* BumpMapping.h:19, -1 = none                    */
    float bump_scale;              /* BumpMapping.h:60                               */
} rtx_material_desc;

/* PointLight (Shading.cpp:65-209): size > 0 is a renderable sphere light. */
typedef struct {
    float position[3];
    float intensity[3];
    float size;
    float attenuation;
} rtx_light_desc;

typedef struct {
    const rtx_mesh_desc* meshes;        int n_meshes;
    const rtx_material_desc* materials; int n_materials;
    const rtx_texmap_desc* texmaps;     int n_texmaps;
    const rtx_light_desc* lights;       int n_lights;
    float environment[3]; int environment_tex;   /* Scene::environment (RayCaster.cpp:368-370) */
    float background[3];  int background_tex;    /* Scene::background  (RayCaster.cpp:367-369) */
    float background_emission;                   /* Scene::backgroundEmission (:371)            */
} rtx_scene_desc;

/* Camera (framework Camera; RayCaster.cpp:717-762, CameraAdapter.h:7-27). dir need not be unit length. */
typedef struct {
    float pos[3], dir[3], up[3];
    float fov;                     /* degrees, vertical */
    float focaldist, dof;
    int width, height;
    int srgb;
} rtx_camera;

/* ---- rays and hits ------------------------------------------------------------------------------------ */
typedef struct {
    float o[3];
    float tmax;                    /* in: HitInfo::z upper bound (IntersectNodeAdapter.h:28,103) */
    float d[3];
    float pad;
} rtx_ray;

/* The fields of the framework HitInfo that the hot path reads. */
typedef struct {
    float t;                       /* HitInfo::z; RTX_BIGFLOAT... unchanged tmax when no hit          */
    int   obj;                     /* index of the (Node,Object) pair, -1 = no hit or light           */
    int   face;                    /* faceID inside that object, -1 = none                            */
    int   mtl_id;                  /* HitInfo::mtlID                                                   */
    float p[3], N[3], GN[3];
    float uvw[3];
    float dpdu[3], dpdv[3];
    float bu, bv;                  /* Moller-Trumbore u,v (objects.cpp:162,166)                       */
    int   front;
    int   light;                   /* index of the hit light (ExtendedHitInfo::hitLight) or -1        */
} rtx_hit;

/* ---- rendering ---------------------------------------------------------------------------------------- */
typedef struct {
    int spp_min;                   /* kMinSamples/kInitSamples (RayCaster.cpp:51-52)                   */
    int spp_max;                   /* kMaxSamples (:53); == spp_min renders a fixed sample count       */
    int index_stride;              /* the 1024 of sampleIndex=(y*W+x)*kMaxSamples+(s+1) (:784); 0 -> 1024 */
    float ci_halfwidth;            /* kMaxHalfWidth (:55)                                              */
    int filter_type;               /* g_filterType (:30)                                               */
    float filter_radius;           /* g_filterRadius (:31); <= 0 selects FilterDefaultSupport; values above
                                      RTX_MAX_FILTER_RADIUS, NaN and unknown filter_type / shading_mode values
                                      are refused with RTX_ERR_INVALID                                  */
    int max_bounces;               /* 64 in RayCaster.cpp:358; 1 = direct lighting only (config C2)    */
    uint64_t seed;                 /* counter-based RNG key                      */
    int tile_size;                 /* bucket edge (RayCaster.cpp:1152), 0 -> 16                        */
    int rank, world;               /* tile sharding: this call renders tiles t with t % world == rank  */
    int samples_per_pass;          /* path-pool sizing: samples per pixel traced per wavefront pass; 0 = auto */
    int use_photons;               /* gather the caustics map when a photon handle is given           */
    int shading_mode;              /* RTX_SHADING_PATH (0, the reference's default build) or RTX_SHADING_LEGACY:
                                    * the SHADING_USE_LEGACY=1 build (ShadingConfig.h:7-13): TraceLegacy
                                    * (RayCaster.cpp:579-631) -> Shade() of Shading_Legacy.cpp through
                                    * ShadeInfoAdapter::TraceShadowRay / TraceSecondaryRay; with a photon handle
                                    * and use_photons it gathers the caustics AND the global map (:434-444)    */
} rtx_render_params;
#define RTX_SHADING_PATH 0
#define RTX_SHADING_LEGACY 1

/* Film accumulators, caller-provided DEVICE memory, all width*height, row-major y*W+x
 * (per-thread buffers of RayCaster.cpp:1168-1171). The caller zeroes accum/count and fills z with
 * RTX_BIGFLOAT (rtx_film_clear does this). Multi-GPU callers sum-reduce accum/count and min-reduce z
 * across ranks (RayCaster.cpp:1182-1212) before rtx_film_resolve; the albedo/normal guides of a pixel are written only
 * by the rank that owns the pixel's bucket (zero elsewhere after rtx_film_clear), so they are sum-reduced as well.
 * The library runs on the null stream of its own HIP runtime: a caller that fills or reduces these buffers with another
 * runtime or stream (torch, RCCL) must synchronise that work with the host before the next rtx_* call that touches them. */
typedef struct {
    float* accum;                  /* float4 per pixel: sum(w*r), sum(w*g), sum(w*b), sum(w)          */
    int*   count;                  /* originSampleCount                                                */
    float* z;                      /* accumZ                                                           */
    float* albedo;                 /* float4 per pixel (first-hit albedo, w=written), optional NULL    */
    float* normal;                 /* float4 per pixel, optional NULL                                  */
} rtx_film;

typedef struct {
    uint64_t rays_primary;
    uint64_t rays_bounce;
    uint64_t rays_shadow;          /* shadow hops (each a closest-hit traversal, RayCaster.cpp:204-274) */
    uint64_t rays_photon;
    uint64_t paths;
    uint64_t nodes_visited;        /* only filled when counters are enabled (rtx_set_counters)         */
    uint64_t tris_tested;
    double   extend_ms;            /* HIP-event time of the extend kernels in the last call             */
    double   shadow_ms;
    double   shade_ms;
    double   total_ms;
    uint64_t extend_launches;
    uint64_t rays_extend;          /* primary + bounce rays traced by the extend launches that extend_ms times
                                    * (bounce rays of the tail / legacy kernels are in rays_bounce only)       */
    /* what the bench prices shade_kernel / nee_kernel / shadow_kernel with (path mode) */
    uint64_t shade_launches;
    uint64_t vertices;             /* path vertices shaded by the shade launches that shade_ms times            */
    uint64_t shadow_launches;
    uint64_t shadow_segments;      /* NEE segments handed to the shadow launches that shadow_ms times           */
    uint64_t shadow_nodes;         /* nodes visited / triangles tested by those launches; only with counters    */
    uint64_t shadow_tris;
    double   nee_ms;               /* the nee_kernel launches alone (also contained in shade_ms)                */
    uint64_t repacks;              /* bounces that moved their paths to consecutive slots (DESIGN.md 3 "Compaction") */
} rtx_stats;

/* Photon mapping (PhotonMapBuilder.h:46-54). */
typedef struct {
    int num_photons;               /* emitted                                                           */
    int max_bounces;               /* 2                                                                 */
    float photon_radius;           /* <= 0: AdjustPhotonGatherSettings (PhotonMapBuilder.cpp:62-78)     */
    float caustic_radius;          /* <= 0: derived                                                     */
    int search_count;              /* 100                                                               */
    uint64_t seed;
    int first_photon, photon_stride; /* multi-GPU split: ids first, first+stride, ... (0,1 = all)       */
} rtx_photon_params;

/* One stored photon: position, direction of travel, power (cyPhotonMap record). */
typedef struct {
    float pos[3];
    float dir[3];
    float power[3];
} rtx_photon;

/* ---- entry points ------------------------------------------------------------------------------------- */

/* Number of HIP devices visible (0 when none). */
int rtx_device_count(void);

/* Selects the HIP device for subsequent calls of this thread (one process per GPU: LOCAL_RANK). */
int rtx_set_device(int device);

const char* rtx_last_error(void);

/* Library version string and build architecture ("gfx950"). */
const char* rtx_version(void);

/* Replaces: Renderer::LoadScene -> TriObj::BuildAcceleration (objects.cpp:128-138, bvh.SetMesh(this,4)) and
 * the scene flattening hook CollectSceneNodeObjects (SceneAdapter.h:8-18). Builds the flattened wide BVH and
 * the compact triangle streams on the host and uploads them. */
int rtx_scene_create(const rtx_scene_desc* desc, rtx_scene** out);
void rtx_scene_destroy(rtx_scene* scene);

/* Scene facts: counts[0]=triangles, [1]=BVH nodes, [2]=node bytes, [3]=triangle record bytes,
 * [4]=device bytes total, [5]=BVH depth. bounds = child-union box of the root (Node::GetChildBoundBox). */
int rtx_scene_info(const rtx_scene* scene, uint64_t counts[6], float bounds[6]);

/* Replaces: IntersectNode(&scene->rootNode, ray, hit, hitSide) (IntersectNodeAdapter.h:100-108) for a batch
 * of rays; with with_lights != 0 it is IntersectSceneWithLights (IntersectNodeAdapter.h:111-151).
 * rays/hits are HOST arrays of n elements. */
int rtx_trace_closest(rtx_scene* scene, const rtx_ray* rays, size_t n, int hit_side, int with_lights,
                      rtx_hit* hits);

/* Same, DEVICE pointers, compact result only: hit4[i] = {t, u, v, bits(prim)}; prim = global triangle index
 * (object order, then face order), -1 = miss, -2-k = light k. stream is a hipStream_t or NULL. Used by the
 * throughput benchmark and by callers that keep rays resident in HBM. */
int rtx_trace_closest_device(rtx_scene* scene, const void* rays_dev, size_t n, int hit_side, int with_lights,
                             void* hit4_dev, void* stream);

/* Replaces: RayCastRenderer::TraceShadowRay (RayCaster.cpp:1412-1422) / Visible() (RayCaster.cpp:178-278):
 * vis[i] in [0,1] for segment origin o, unit direction d, length tmax; use_transmittance selects the
 * 8-hop Fresnel chain (:199-277) or the boolean test (:178-190). Host arrays. */
int rtx_trace_visibility(rtx_scene* scene, const rtx_ray* rays, const float* normals, size_t n,
                         int use_transmittance, float* vis);

/* Replaces: GenerateCameraRay (CameraAdapter.h:7-27) + IntersectNode per pixel (config C1): fills host
 * arrays face[W*H] (global triangle index or -1), t[W*H], front[W*H] (may be NULL). */
int rtx_render_primary(rtx_scene* scene, const rtx_camera* cam, int hit_side, int* prim, float* t,
                       unsigned char* front);

/* The ShadeInfo state that ShadeInfoAdapter's ray queries read (ShadeInfoAdapter.h:40-47, :60, :75, :105):
 * hit point, shading normal, receiver node, recursion depths, and the pixel-sample key of its random numbers. */
typedef struct {
    float P[3], N[3];
    int obj;                       /* GetNode(): index of the receiver's (Node,Object) pair, -1 = none          */
    int bounce;                    /* ShadeInfo::bounce (CanBounce: bounce < 5)                                  */
    int diffuse_depth;             /* ShadeInfoAdapter::diffuseDepth                                             */
    int pixel, sample;             /* SetPixel (y*W+x) / SetPixelSample: key of the counter RNG                  */
    unsigned int node;             /* this Shade() in the recursion tree: 0 = primary hit, child k of node m =
                                    * 3m+k (k = 1 diffuse, 2 reflection, 3 refraction); keys the RNG stream      */
} rtx_shade_point;

/* Replaces: ShadeInfoAdapter::TraceSecondaryRay(Ray(P, dirs[i]), dist, recursive, isReflection)
 * (ShadeInfoAdapter.h:94-231; legacy shading mode) for n directions from one shading point: biased origin,
 * nearest of renderable lights and geometry, Shade() of the hit (Shading_Legacy.cpp) with its own shadow and
 * secondary rays. `child` (1..3) names which call of the caller's Shade() this is (RNG node of the hit);
 * photons may be NULL. color_dist[4i..4i+3] = {r, g, b, dist}. Host arrays. */
int rtx_trace_secondary(rtx_scene* scene, const rtx_photons* photons, const rtx_shade_point* at, const float* dirs, size_t n,
                        int recursive, int child, uint64_t seed, float* color_dist);

/* Film helpers (device memory). */
int rtx_film_clear(const rtx_film* film, int width, int height, void* stream);

/* Replaces: RayCaster::CastRays (RayCaster.cpp:1146-1314) minus the merge: traces all samples of the tiles
 * owned by params->rank and accumulates into film (device). photons may be NULL. Blocking. */
int rtx_render_accumulate(rtx_scene* scene, const rtx_camera* cam, const rtx_render_params* params,
                          rtx_photons* photons, const rtx_film* film, rtx_stats* stats);

/* Replaces: mergeToImage (RayCaster.cpp:1182-1212) + ToColor24 (:33-48): final = sum/weight, 8-bit
 * quantisation. Outputs are HOST arrays (any may be NULL): linear float3, rgb8, z, samples. */
int rtx_film_resolve(const rtx_film* film, const rtx_camera* cam, float* linear_rgb, unsigned char* rgb8,
                     float* z, int* samples, float* albedo, float* normal);

/* Per-pixel running statistics of the adaptive sampler (RayCaster.cpp:832-838): device buffers of width*height
 * float4: sum = {sum r, g, b, bits(n)}, sum2 = {sum r^2, g^2, b^2, running z}. All-zero is the empty state. */
typedef struct {
    float* sum;
    float* sum2;
} rtx_pixel_stats;

/* Replaces: Denoiser::ApplyMedian / MedianFilterToLinear (Denoiser.cpp:19-134): 3x3 median per channel with the
 * window clamped to the image, then clamp to [0,1], optional Linear2sRGB and truncation to 8 bits. Host arrays:
 * linear_rgb_in width*height*3; outputs optional (linear_rgb_out width*height*3, rgb8_out width*height*3). */
int rtx_median_denoise(const float* linear_rgb_in, int width, int height, int srgb, float* linear_rgb_out,
                       unsigned char* rgb8_out);

/* Film and statistics buffers owned by the library, for hosts that do not manage device memory themselves (the C++
 * mirror of RayCaster's progressive mode). Cleared on creation; albedo/normal guides included. */
int rtx_film_create(int width, int height, rtx_film* film, rtx_pixel_stats* stats);
void rtx_film_destroy(rtx_film* film, rtx_pixel_stats* stats);

/* One sampling round of RenderBucket (RayCaster.cpp:770-840, :896-956) on an explicit pixel list: samples
 * [sample_base, sample_base + n_samples) of every listed pixel (index y*width+x, host array, any order) are traced
 * and splatted into `film`, and the pixels' statistics are updated. reset_z restarts the running z as the reference
 * does when refinement begins (:905). Building blocks of (a) multi-GPU adaptive sampling with per-round
 * re-balancing of tiles (raytracer_amd/distributed.py) and (b) progressive 1-spp passes
 * (RenderProgressiveBucket, RayCaster.cpp:1025-1144). params->spp_min/spp_max/rank/world are ignored. */
int rtx_render_round(rtx_scene* scene, const rtx_camera* cam, const rtx_render_params* params, rtx_photons* photons,
                     const rtx_film* film, const rtx_pixel_stats* stats, const int* pixels, size_t n_pixels, int sample_base,
                     int n_samples, int reset_z, rtx_stats* out_stats);

/* The confidence-interval test of RayCaster.cpp:848-868 over the whole frame: pixels (ascending y*width+x) with
 * 1 < n < spp_max whose half-width exceeds ci_halfwidth. Returns their number; writes up to `capacity` of them. */
int64_t rtx_refine_list(const rtx_pixel_stats* stats, int width, int height, float ci_halfwidth, int spp_max, int* out_pixels,
                        size_t capacity);

/* No counterpart in the reference. rtx_refine_list for callers that cannot allocate device memory: code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc. Same test, same ascending order, but nothing is
 * allocated, freed or synchronised: the kernels are enqueued on `stream` (a hipStream_t or NULL) and the results go to
 * caller-owned DEVICE buffers: out_pixels_dev[0 .. min(count, capacity)) and *out_count_dev = count. The scratch of
 * the ordered compaction is a static device array inside the library (about 256 KB per GPU, held until the process
 * ends), which bounds the frame: width*height above RTX_REFINE_NOALLOC_MAX_PIXELS is refused with RTX_ERR_TOO_LARGE.
 * All calls on one GPU share that array: calls that overlap on one GPU race (serialise them on one stream).
 * The first call on a device looks the array's address up (hipGetSymbolAddress, which may load the library's code object):
 * make one call per device outside any graph capture before capturing it. */
#define RTX_REFINE_NOALLOC_MAX_PIXELS (1ull << 26)
int rtx_refine_list_noalloc(const rtx_pixel_stats* stats, int width, int height, float ci_halfwidth, int spp_max,
                            int* out_pixels_dev, size_t capacity, uint32_t* out_count_dev, void* stream);

/* The library runs on its own HIP runtime: a host that enqueues rtx_* work asynchronously - the entry point
 * above, rtx_film_clear with a stream - waits for it with these, not with another runtime's synchronisation (torch's sees
 * only torch's streams). rtx_device_synchronize: everything enqueued on the current device. rtx_stream_*: a stream of the
 * library's runtime, usable wherever an rtx_* entry point takes `void* stream`. No counterpart in the reference. */
int rtx_device_synchronize(void);
int rtx_stream_create(void** out_stream);
int rtx_stream_synchronize(void* stream);
int rtx_stream_destroy(void* stream);

/* Replaces: RayCaster::Stop (RayCaster.cpp:1316-1318): asks an in-flight rtx_render* call on this scene (running in
 * another thread) to stop after the current bounce; that call returns RTX_OK with the samples accumulated so far.
 * The flag is cleared when the next render call starts. Async-safe. */
int rtx_cancel(rtx_scene* scene);

/* Convenience single-GPU CastRays: clear + accumulate + resolve. */
int rtx_render(rtx_scene* scene, const rtx_camera* cam, const rtx_render_params* params, rtx_photons* photons,
               float* linear_rgb, unsigned char* rgb8, float* z, int* samples, rtx_stats* stats);

/* Debug / parity aid with no counterpart in the reference: per-sample results of a fixed-spp render
 * (spp_max == spp_min). out[(y*W+x)*spp + s] = {r, g, b (after the radiance clamp), primaryDist}, HOST array. */
int rtx_render_samples(rtx_scene* scene, const rtx_camera* cam, const rtx_render_params* params, rtx_photons* photons,
                       float* out, rtx_stats* stats);

/* Replaces: PhotonMapBuilder::BuildPhotonMap (PhotonMapBuilder.cpp:20-60): emission + tracing on the GPU.
 * The stored photons stay in HBM inside the handle. */
int rtx_photon_build(rtx_scene* scene, const rtx_photon_params* params, rtx_photons** out, rtx_stats* stats);
void rtx_photons_destroy(rtx_photons* p);

/* counts[0]=global photons stored, [1]=caustic photons stored; radii[0]=photon radius, [1]=caustic radius. */
int rtx_photon_info(const rtx_photons* p, uint64_t counts[2], float radii[2]);

/* Copies stored photon records (unscaled powers, emission order not guaranteed) to host. which: 0 global,
 * 1 caustic. Returns number copied or negative error. */
int64_t rtx_photon_download(const rtx_photons* p, int which, rtx_photon* out, size_t capacity);

/* Multi-GPU: replace the caustic/global photon set with the concatenation gathered from all ranks (device
 * pointer to n raw records) and rebuild the hash grid + power scaling (PhotonMapBuilder.cpp:38-50).
 * Raw record = 12 floats: {pos.xyz, bits(photonId*4+bounce)}, {dir.xyz, 0}, {unscaled power.rgb, 0}.
 * The reference's map capacities apply (PhotonMapBuilder.cpp:27-28: total_emitted records in the global map,
 * total_emitted / 10 in the caustics map; a full map takes no more photons): beyond them the records with the smallest
 * (photon, bounce) keys stay, here and in rtx_photon_build - the same map from one GPU or from many. */
int rtx_photon_set(rtx_photons* p, int which, const void* photons_dev, size_t n, int total_emitted);

/* Device address of the stored raw records (48 B each, layout above) for an RCCL all-gather by the caller. */
void* rtx_photon_device_ptr(const rtx_photons* p, int which);

/* Copies the stored raw records of set `which` (48 B each, layout above) into a caller-owned DEVICE buffer of
 * `capacity` records: the send buffer of the all-gather (raytracer_amd/distributed.py: all_gather_photons; all workers of
 * PhotonMapBuilder.cpp:102-146 feed one map). Synchronous. Returns the number of records copied or a negative error. */
int64_t rtx_photon_copy_raw(const rtx_photons* p, int which, void* dst_dev, size_t capacity);

/* Replaces: EstimateCaustics / PhotonMap::EstimateIrradiance<N,LINEAR> (RayCaster.cpp:143-153) for host
 * probe points: irradiance rgb per probe. */
int rtx_photon_estimate(rtx_photons* p, int which, const float* positions, const float* normals, size_t n,
                        float* irradiance);

/* ---- several GPUs, one host process ---------------------------------------------------------------------------
 * Replaces: the worker pool of RayCaster::CastRays (RayCaster.cpp:1281-1312: nThreads workers draw 16x16 buckets) and
 * mergeToImage (:1182-1212) across the GPUs of one node, for hosts that stay one C++ process (raytracer_amd/host/
 * RayCaster.h selects it through SamplingConfig::devices). A group holds one member per entry of device_ids, each with
 * its own replica of the scene in that GPU's HBM; an id may repeat (members then share the GPU: how the one-GPU tests
 * exercise the protocol). rtx_group_render runs one host thread per member: member m traces the buckets b with
 * b % n == m into a full-frame film on its GPU (params->rank / world are ignored); the films are then copied to member
 * 0's GPU (hipMemcpyPeer: direct xGMI transfers from all peers at once) and merged there in member order - colour,
 * weight, counts and guides by sum, z by min - and resolved into the host arrays (any may be NULL), as rtx_render does.
 * Adaptive sampling (spp_max > spp_min): round 1 on that static split; after every round the statistics of the pixels
 * each member sampled are collected on member 0's GPU, the confidence-interval test (RayCaster.cpp:848-868) runs there,
 * and the buckets that still hold unconverged pixels are dealt out again, heaviest first to the least loaded member
 * (SURVEY.md 8e); only statistics move, accumulated colour stays in the film of the member that traced it. Sample maps
 * are those of one GPU. The one-process-per-GPU alternative (RCCL) is raytracer_amd/distributed.py. */
#define RTX_GROUP_MAX 16
typedef struct rtx_group rtx_group;
int rtx_group_create(const rtx_scene_desc* desc, const int* device_ids, int n, rtx_group** out);
void rtx_group_destroy(rtx_group* g);
int rtx_group_size(const rtx_group* g);
/* The scene replica / photon maps of one member (borrowed; for ray queries: rtx_set_device(its device) first). */
rtx_scene* rtx_group_scene(rtx_group* g, int member);
rtx_photons* rtx_group_photons(rtx_group* g, int member);
int rtx_group_render(rtx_group* g, const rtx_camera* cam, const rtx_render_params* params, float* linear_rgb,
                     unsigned char* rgb8, float* z, int* samples, float* albedo, float* normal, rtx_stats* stats);
/* Replaces: PhotonMapBuilder::BuildPhotonMap with its worker threads feeding one map (PhotonMapBuilder.cpp:20-60,
 * :102-146): member m traces the photon ids m, m + n, ...; the raw records of all members are then gathered onto every
 * member's GPU (peer copies) and each member rebuilds its hash grid and power scaling from the whole set
 * (params->first_photon / photon_stride are ignored). The maps stay inside the group; rtx_group_render gathers from
 * them when params->use_photons is set. stats: rays_photon summed over members. */
int rtx_group_photon_build(rtx_group* g, const rtx_photon_params* params, rtx_stats* stats);
/* RayCaster::Stop for a group render in flight (async-safe). */
int rtx_group_cancel(rtx_group* g);

/* Enables per-ray node/triangle counters (instrumented traversal; slower) and zeroes them. */
int rtx_set_counters(rtx_scene* scene, int enabled);

/* out[0]=BVH nodes visited, [1]=triangles tested, [2]=rays traced since rtx_set_counters(scene, 1). These give
 * the algorithmic bytes per ray of SURVEY.md §8(d): 48 + nodes*64 + tris*48. */
int rtx_get_counters(rtx_scene* scene, uint64_t out[3]);

/* Host-only checks (no device needed; used by the CPU test-suite). No counterpart in the reference.
 * rtx_debug_validate_bvh: flattens desc and verifies the BVH (every triangle in exactly one leaf, quantised boxes
 * conservative); out[0]=triangles, [1]=nodes, [2]=depth.
 * rtx_debug_pixel_list: the padded pixel list (64-pixel wave blocks, -1 = padding) of the buckets that
 * RayCaster.cpp:1152-1158 would build and that belong to `rank` of `world`; returns its length.
 * rtx_debug_lpt_assign: the deal rtx_group_render makes between adaptive rounds (greedy longest-processing-time: buckets by
 * decreasing weight then index, each to the least loaded member, ties to the lowest): owner[b] in [0, members) for buckets
 * with weight > 0, -1 otherwise. The same rule as raytracer_amd/distributed.py::lpt_assign.
 * rtx_debug_queue_layout: geometry of the segmented ray queues (csrc/segqueue.h) for a producer grid of
 * `producer_blocks` blocks of `block_threads` threads appending at most `mult` entries per thread: start[s] = first
 * item index of segment s (64 segments), segment_of_block[b] for every block (array of producer_blocks).
 * rtx_debug_queue_layout_run: the same for a queue whose producer blocks go round the segments in runs of `run` blocks
 * (SegQ::run, the pair queue between shade_kernel and nee_kernel; run = 0: contiguous segments, as above). */
int rtx_debug_validate_bvh(const rtx_scene_desc* desc, uint64_t out[3]);
/* A host model of the kernels' traversal order (no device needed): flattens desc, walks n rays {o, tmax, d} to their closest
 * hit among the world-space triangles in the order the kernels visit the tree, and returns t_out[i] (-1 = miss), prim_out[i]
 * (global triangle id, -1 = miss), counts[0] = nodes visited, [1] = triangles tested, [2] = nodes of the tree, [3] = depth.
 * The CPU tests hold it to a brute-force loop (the tree loses no hit), and it prices a change of the tree layout in steps
 * per ray before any kernel is written. Not a render path. */
int rtx_debug_host_walk(const rtx_scene_desc* desc, const rtx_ray* rays, size_t n, float* t_out, int* prim_out, uint64_t counts[4]);
/* The elementary functions of the path as this library evaluates them (raytracer_amd/csrc/rtx_math.h; the reference calls
 * the platform libm): fn 0 powf(x, y), 1 expf(x), 2 sinf, 3 cosf, 4 tanf, 5 asinf, 6 acosf, 7 atan2f(x, y), 8 x / 255 for
 * x = 0 ... 255 (the texel conversion, which must equal the IEEE division); host arrays
 * of n values. on_device = 0 evaluates on the host side of the library (no GPU needed), 1 inside a kernel: the tests hold
 * both, bit for bit, to the CPU oracle's evaluation of the same source. */
int rtx_debug_math(int fn, const float* x, const float* y, size_t n, float* out, int on_device);
int rtx_debug_queue_layout(uint32_t producer_blocks, uint32_t block_threads, uint32_t mult, uint32_t start[64],
                           uint32_t* segment_of_block);
int rtx_debug_queue_layout_run(uint32_t producer_blocks, uint32_t block_threads, uint32_t mult, uint32_t run, uint32_t start[64],
                               uint32_t* segment_of_block);
int64_t rtx_debug_pixel_list(int width, int height, int tile, int rank, int world, int* out, size_t capacity);
int rtx_debug_lpt_assign(const int* weight, size_t n_buckets, int members, int* owner);

#ifdef __cplusplus
}
#endif
#endif /* RTX_H */
