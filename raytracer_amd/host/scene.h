// Host-side scene types for the C++ mirror of the reference's hot-path interface.
//
// The reference compiles against a course framework (scene.h, objects.h, materials.h, lights.h, renderer.h, cy*.h)
// that is absent from its repository. These are NOT those files: they are the minimal plain types the
// mirrored entry points (RayCaster, IntersectNode, IntersectSceneWithLights, CollectSceneNodeObjects,
// ShadeInfoAdapter, PhotonMapBuilder) need in their signatures, with the member names the reference's call sites
// use. All ray casting and light transport happens on the GPU behind the C ABI (include/rtx.h); nothing here
// intersects or shades anything.
#pragma once
#include <cmath>
#include <cstdint>
#include <memory>
#include <string>
#include <utility>
#include <vector>
#include "../../include/rtx.h"
#include "ImageIO.h"

namespace rtamd {

constexpr float BIGFLOAT = RTX_BIGFLOAT;
enum { HIT_NONE = 0, HIT_FRONT = 1, HIT_BACK = 2, HIT_FRONT_AND_BACK = 3 };

struct Vec3f {
    float x = 0, y = 0, z = 0;
    Vec3f() = default;
    Vec3f(float a, float b, float c) : x(a), y(b), z(c) {}
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    Vec3f operator+(Vec3f const& o) const { return {x + o.x, y + o.y, z + o.z}; }
    Vec3f operator-(Vec3f const& o) const { return {x - o.x, y - o.y, z - o.z}; }
    Vec3f operator-() const { return {-x, -y, -z}; }
    Vec3f operator*(float s) const { return {x * s, y * s, z * s}; }
    Vec3f operator/(float s) const { return {x / s, y / s, z / s}; }
    float operator%(Vec3f const& o) const { return x * o.x + y * o.y + z * o.z; }                      // dot
    Vec3f operator^(Vec3f const& o) const { return {y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x}; }  // cross
    float Dot(Vec3f const& o) const { return *this % o; }
    Vec3f Cross(Vec3f const& o) const { return *this ^ o; }
    float LengthSquared() const { return *this % *this; }
    float Length() const { return std::sqrt(LengthSquared()); }
    void Normalize() { *this = *this / Length(); }
    Vec3f GetNormalized() const { return *this / Length(); }
    void Set(float a, float b, float c) { x = a; y = b; z = c; }
    void Zero() { x = y = z = 0; }
};

struct Color {
    float r = 0, g = 0, b = 0;
    Color() = default;
    explicit Color(float v) : r(v), g(v), b(v) {}
    Color(float R, float G, float B) : r(R), g(G), b(B) {}
};
struct Color24 {
    uint8_t r = 0, g = 0, b = 0;
};

// 3x4 affine transform (row-major), identity by default
struct Matrix34 {
    float m[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    static Matrix34 Compose(Matrix34 const& a, Matrix34 const& b) {   // a * b
        Matrix34 r;
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) r.m[i * 4 + j] = a.m[i * 4] * b.m[j] + a.m[i * 4 + 1] * b.m[4 + j] + a.m[i * 4 + 2] * b.m[8 + j];
            r.m[i * 4 + 3] = a.m[i * 4] * b.m[3] + a.m[i * 4 + 1] * b.m[7] + a.m[i * 4 + 2] * b.m[11] + a.m[i * 4 + 3];
        }
        return r;
    }
};

struct Ray {
    Vec3f p, dir;
    Ray() = default;
    Ray(Vec3f const& P, Vec3f const& D) : p(P), dir(D) {}
};

struct Box {
    Vec3f pmin{BIGFLOAT, BIGFLOAT, BIGFLOAT}, pmax{-BIGFLOAT, -BIGFLOAT, -BIGFLOAT};
    Box() = default;
    Box(Vec3f lo, Vec3f hi) : pmin(lo), pmax(hi) {}
    bool IsEmpty() const { return pmin.x > pmax.x || pmin.y > pmax.y || pmin.z > pmax.z; }
};

class Node;
class Light;

struct HitInfo {
    float z;
    Node const* node;
    Vec3f p, N, GN, uvw, duvw[2], dpdu, dpdv;
    int mtlID;
    bool front, light;
    HitInfo() { Init(); }
    void Init() {
        z = BIGFLOAT;
        node = nullptr;
        p = N = GN = dpdu = dpdv = Vec3f();
        uvw = Vec3f(0.5f, 0.5f, 0.5f);
        duvw[0].Zero();
        duvw[1].Zero();
        mtlID = 0;
        front = true;
        light = false;
    }
};

// ---- textures / materials / lights: parameter holders that are flattened into the rtx_scene_desc ----
struct TextureMap {
    int width = 0, height = 0;
    bool isFloat = false;
    std::vector<uint8_t> rgb8;     // width*height*3
    std::vector<float> rgb32f;     // width*height*3
    Matrix34 itm;                  // texture-space transform applied to uvw before lookup
};
struct TexturedColor {
    Color value{0, 0, 0};
    std::shared_ptr<TextureMap> map;
    Color GetValue() const { return value; }
    TextureMap const* GetTexture() const { return map.get(); }
};

class Material {
public:
    virtual ~Material() = default;
    virtual int Type() const = 0;
    TexturedColor diffuse, specular, reflection, refraction, emission;
    float glossiness = 20.0f, ior = 1.5f;
    std::shared_ptr<TextureMap> glossinessMap;   // TexturedFloat Glossiness(): value * red channel of the map (Shading.cpp:238)
    Color absorption{0, 0, 0};
    std::shared_ptr<TextureMap> bump;
    float bumpIntensity = 1.0f;
    TexturedColor const& Diffuse() const { return diffuse; }
    TexturedColor const& Refraction() const { return refraction; }
    float IOR(int = 0) const { return ior; }
    Color Absorption(int = 0) const { return absorption; }
    bool IsPhotonSurface(int = 0) const { return diffuse.value.r + diffuse.value.g + diffuse.value.b > 0.0f; }
};
class MtlBlinn : public Material {
public:
    MtlBlinn() { diffuse.value = Color(0.5f); specular.value = Color(0.7f); }
    int Type() const override { return RTX_MTL_BLINN; }
};
class MtlPhong : public Material {
public:
    MtlPhong() { diffuse.value = Color(0.5f); specular.value = Color(0.7f); }
    int Type() const override { return RTX_MTL_PHONG; }
};

class Light {
public:
    virtual ~Light() = default;
    virtual bool IsRenderable() const { return false; }
    virtual bool IsPhotonSource() const { return false; }
};
class PointLight : public Light {
public:
    Vec3f position;
    Color intensity{1, 1, 1};
    float size = 0.0f, attenuation = 1.0f;
    bool IsRenderable() const override { return size > 0.0f; }
    bool IsPhotonSource() const override { return true; }
    Color Intensity() const { return intensity; }
};

// ---- objects and the node tree ----
class Object {
public:
    virtual ~Object() = default;
    virtual Box GetBoundBox() const = 0;
};

// Unit sphere at the origin of object space (objects.cpp:21-86) and the square [-1,1]^2 at z = 0 (objects.cpp:89-126);
// both are intersected on the GPU in object space under their node's transform.
class Sphere : public Object {
public:
    Box GetBoundBox() const override { return Box(Vec3f(-1, -1, -1), Vec3f(1, 1, 1)); }
};
class Plane : public Object {
public:
    Box GetBoundBox() const override { return Box(Vec3f(-1, -1, 0), Vec3f(1, 1, 0)); }
};

// Triangle mesh with cy::TriMesh's accessors (V/VN/VT, F/FN/FT); BuildAcceleration only records the bounds: the
// acceleration structure is built when the scene is flattened for the GPU (objects.cpp:128-138).
class TriObj : public Object {
public:
    struct TriFace {
        int v[3];
    };
    std::vector<Vec3f> v, vn, vt;
    std::vector<TriFace> f, fn, ft;
    std::vector<int> faceMtl;
    unsigned int NF() const { return (unsigned int)f.size(); }
    bool HasNormals() const { return !vn.empty(); }
    bool HasTextureVertices() const { return !vt.empty(); }
    Vec3f const& V(int i) const { return v[i]; }
    TriFace const& F(int i) const { return f[i]; }
    int GetMaterialIndex(int face) const { return faceMtl.empty() ? 0 : faceMtl[face]; }
    void BuildAcceleration() {
        bounds = Box();
        for (auto const& p : v) {
            bounds.pmin = Vec3f(std::fmin(bounds.pmin.x, p.x), std::fmin(bounds.pmin.y, p.y), std::fmin(bounds.pmin.z, p.z));
            bounds.pmax = Vec3f(std::fmax(bounds.pmax.x, p.x), std::fmax(bounds.pmax.y, p.y), std::fmax(bounds.pmax.z, p.z));
        }
    }
    Box GetBoundBox() const override { return bounds; }

private:
    Box bounds;
};

class Node {
public:
    Matrix34 tm;   // node -> parent
    void AppendChild(Node* n) { children.emplace_back(n); }
    int GetNumChild() const { return (int)children.size(); }
    Node const* GetChild(int i) const { return children[i].get(); }
    Node* GetChild(int i) { return children[i].get(); }
    Object const* GetNodeObj() const { return obj.get(); }
    void SetNodeObj(std::shared_ptr<Object> o) { obj = std::move(o); }
    Material const* GetMaterial() const { return mtl.get(); }
    void SetMaterial(std::shared_ptr<Material> m) { mtl = std::move(m); }

private:
    std::vector<std::unique_ptr<Node>> children;
    std::shared_ptr<Object> obj;
    std::shared_ptr<Material> mtl;
};

struct Scene {
    Node rootNode;
    std::vector<Light*> lights;          // borrowed by the ray caster, owned here
    std::vector<std::unique_ptr<Light>> lightStorage;
    TexturedColor environment, background;
    float backgroundEmission = 0.0f;
    PointLight* AddPointLight() {
        lightStorage.emplace_back(new PointLight());
        lights.push_back(lightStorage.back().get());
        return static_cast<PointLight*>(lights.back());
    }
};

struct Camera {
    Vec3f pos{0, 0, 0}, dir{0, 0, -1}, up{0, 1, 0};
    float fov = 40.0f, focaldist = 1.0f, dof = 0.0f;
    int imgWidth = 200, imgHeight = 150;
    bool sRGB = false;

    // Camera animation (assets/cerealBar.xml:220-271 <keyframe time> with position / target / fov; consumed by the frame
    // loop of Main.cpp:381-401 through Camera::EvaluateAtTime). The framework's interpolation is not in the reference
    // repository: piecewise linear in position, target and fov here, held constant outside the keyed range.
    struct Keyframe {
        double time;
        Vec3f pos, target;
        float fov;
    };
    std::vector<Keyframe> keyframes;   // sorted by time
    Vec3f upHint{0, 1, 0};             // the <up> vector the orthonormal frame is rebuilt from
    bool IsAnimated() const { return !keyframes.empty(); }
    double GetAnimationStart() const { return keyframes.empty() ? 0.0 : keyframes.front().time; }
    double GetAnimationEnd() const { return keyframes.empty() ? 0.0 : keyframes.back().time; }
    void LookAt(Vec3f const& p, Vec3f const& target, bool keepFocalDist = true) {
        pos = p;
        const Vec3f d = target - p;
        if (!keepFocalDist) focaldist = d.Length();
        dir = d.GetNormalized();
        const Vec3f x = dir ^ upHint;
        up = (x ^ dir).GetNormalized();
    }
    void EvaluateAtTime(double t) {
        if (keyframes.empty()) return;
        if (t <= keyframes.front().time || keyframes.size() == 1) {
            fov = keyframes.front().fov;
            LookAt(keyframes.front().pos, keyframes.front().target);
            return;
        }
        if (t >= keyframes.back().time) {
            fov = keyframes.back().fov;
            LookAt(keyframes.back().pos, keyframes.back().target);
            return;
        }
        size_t k = 1;
        while (keyframes[k].time < t) ++k;
        const Keyframe &a = keyframes[k - 1], &b = keyframes[k];
        const float w = (float)((t - a.time) / (b.time - a.time));
        fov = a.fov + (b.fov - a.fov) * w;
        LookAt(a.pos + (b.pos - a.pos) * w, a.target + (b.target - a.target) * w);
    }
};

// RenderImage: the output arrays CastRays fills (RayCaster.cpp:1000-1012, 1183-1211), row-major y*W+x
class RenderImage {
public:
    void Init(int w, int h) {
        width = w;
        height = h;
        img.assign((size_t)w * h, Color24());
        zbuffer.assign((size_t)w * h, BIGFLOAT);
        sampleCount.assign((size_t)w * h, 0);
        linear.assign((size_t)w * h, Color());
        albedo.assign((size_t)w * h, Color());
        normal.assign((size_t)w * h, Vec3f());
        numRendered = 0;
    }
    int GetWidth() const { return width; }
    int GetHeight() const { return height; }
    Color24* GetPixels() { return img.data(); }
    float* GetZBuffer() { return zbuffer.data(); }
    int* GetSampleCount() { return sampleCount.data(); }
    Color* GetLinearPixels() { return linear.data(); }
    Color* GetAlbedo() { return albedo.data(); }
    Vec3f* GetNormal() { return normal.data(); }
    // 8-bit views of the z buffer and the sample counts, as the framework's RenderImage computes them before saving
    // (stand-in for the absent framework code: nearest = white, background (BIGFLOAT) = black; counts scaled min..max).
    void ComputeZBufferImage() {
        zImg.assign(zbuffer.size(), 0);
        float zmin = BIGFLOAT, zmax = 0.0f;
        for (float z : zbuffer) {
            if (z == BIGFLOAT) continue;
            zmin = std::fmin(zmin, z);
            zmax = std::fmax(zmax, z);
        }
        for (size_t i = 0; i < zbuffer.size(); ++i) {
            if (zbuffer[i] == BIGFLOAT) continue;
            const float f = zmax > zmin ? (zmax - zbuffer[i]) / (zmax - zmin) : 1.0f;
            int c = (int)(f * 255.0f);
            zImg[i] = (uint8_t)(c < 0 ? 0 : (c > 255 ? 255 : c));
        }
    }
    int ComputeSampleCountImage() {
        sImg.assign(sampleCount.size(), 0);
        int smin = 0x7FFFFFFF, smax = 0;
        for (int s : sampleCount) {
            smin = s < smin ? s : smin;
            smax = s > smax ? s : smax;
        }
        if (smax > smin)
            for (size_t i = 0; i < sampleCount.size(); ++i) sImg[i] = (uint8_t)((255 * (sampleCount[i] - smin)) / (smax - smin));
        return smax;
    }
    uint8_t* GetZBufferImage() { return zImg.data(); }
    uint8_t* GetSampleCountImage() { return sImg.data(); }
    bool SaveImage(const char* filename) const { return png::save(filename, &img.data()->r, width, height, 3); }
    bool SaveZImage(const char* filename) const { return zImg.size() == img.size() && png::save(filename, zImg.data(), width, height, 1); }
    bool SaveSampleCountImage(const char* filename) const { return sImg.size() == img.size() && png::save(filename, sImg.data(), width, height, 1); }
    void IncrementNumRenderPixel(int n) { numRendered += n; }
    int GetNumRenderedPixels() const { return numRendered; }
    void ResetNumRenderedPixels() { numRendered = 0; }

private:
    int width = 0, height = 0, numRendered = 0;
    std::vector<Color24> img;
    std::vector<float> zbuffer;
    std::vector<int> sampleCount;
    std::vector<Color> linear, albedo;
    std::vector<Vec3f> normal;
    std::vector<uint8_t> zImg, sImg;
};

}  // namespace rtamd
