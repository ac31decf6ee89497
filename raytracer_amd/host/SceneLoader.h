// Scene ingest for the C++ host mirror: the XML dialect of assets/cerealBar.xml and Wavefront OBJ
// meshes, read into the scene types of scene.h so that RayCaster / GpuScene take them unchanged.
//
// The reference reads scenes through the course framework's xmlload + the vendored tinyxml2 and cyTriMesh, none of
// which are part of the hot path (and the framework files are absent from the reference repository). This is a
// self-contained reader of the subset those files use:
//   <object type="obj|sphere|plane" name=... material=...> with nested <scale>, <rotate>, <translate> and child
//       <object>s; transforms compose in document order (each one is applied after the ones before it)
//   <material type="blinn|phong" name=...> with <diffuse|specular|reflection|refraction|emission r g b value texture>,
//       <glossiness value>, <refraction index>, <absorption r g b>, <bump texture value>
//   <environment|background r g b value texture> with nested texture transforms, <light type="point"> with
//       <intensity>, <position>, <size>, <attenuation>, and <camera gamma="sRGB"> with position/target/up/fov/
//       focaldist/dof/width/height and <keyframe time> children (position/target/fov) for Camera::EvaluateAtTime
// A colour is (r, g, b) (default 1) times `value` (default 1). Texture files are resolved by a caller-supplied
// function first; binary PPM (P6), PFM (PF), PNG and Radiance HDR (ImageDecode.h: the two formats the reference's
// scene file names) are read directly.
#pragma once
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>
#include <sstream>
#include "scene.h"
#include "ImageDecode.h"

namespace rtamd {

// ---- a small XML reader: elements, attributes, comments, declarations; text content is ignored ----
struct XmlNode {
    std::string name;
    std::map<std::string, std::string> attr;
    std::vector<XmlNode> children;
    const XmlNode* Child(const char* n) const {
        for (auto const& c : children)
            if (c.name == n) return &c;
        return nullptr;
    }
    bool Has(const char* k) const { return attr.count(k) != 0; }
    std::string Str(const char* k, const std::string& def = "") const {
        auto it = attr.find(k);
        return it == attr.end() ? def : it->second;
    }
    float Float(const char* k, float def) const {
        auto it = attr.find(k);
        return it == attr.end() ? def : (float)std::atof(it->second.c_str());
    }
};

class XmlParser {
public:
    explicit XmlParser(const std::string& text) : s(text) {}
    bool Parse(XmlNode& root, std::string& err) {
        root = XmlNode();
        root.name = "#document";
        std::vector<XmlNode*> stack{&root};
        while (true) {
            size_t lt = s.find('<', pos);
            if (lt == std::string::npos) break;
            pos = lt;
            if (s.compare(pos, 4, "<!--") == 0) {
                size_t e = s.find("-->", pos + 4);
                if (e == std::string::npos) return Fail(err, "unterminated comment");
                pos = e + 3;
                continue;
            }
            if (s.compare(pos, 2, "<?") == 0 || s.compare(pos, 2, "<!") == 0) {
                size_t e = s.find('>', pos);
                if (e == std::string::npos) return Fail(err, "unterminated declaration");
                pos = e + 1;
                continue;
            }
            if (s.compare(pos, 2, "</") == 0) {
                size_t e = s.find('>', pos);
                if (e == std::string::npos) return Fail(err, "unterminated end tag");
                std::string n = Trim(s.substr(pos + 2, e - pos - 2));
                if (stack.size() < 2 || stack.back()->name != n) return Fail(err, "mismatched </" + n + ">");
                stack.pop_back();
                pos = e + 1;
                continue;
            }
            ++pos;
            XmlNode node;
            node.name = ReadName();
            if (node.name.empty()) return Fail(err, "element name expected");
            bool selfClose = false;
            while (true) {
                SkipSpace();
                if (pos >= s.size()) return Fail(err, "unterminated tag <" + node.name + ">");
                if (s[pos] == '/') {
                    selfClose = true;
                    ++pos;
                    continue;
                }
                if (s[pos] == '>') {
                    ++pos;
                    break;
                }
                std::string k = ReadName();
                SkipSpace();
                if (k.empty() || pos >= s.size() || s[pos] != '=') return Fail(err, "attribute syntax in <" + node.name + ">");
                ++pos;
                SkipSpace();
                if (pos >= s.size() || (s[pos] != '"' && s[pos] != '\'')) return Fail(err, "quoted attribute value expected");
                const char q = s[pos++];
                size_t e = s.find(q, pos);
                if (e == std::string::npos) return Fail(err, "unterminated attribute value");
                node.attr[k] = s.substr(pos, e - pos);
                pos = e + 1;
            }
            stack.back()->children.push_back(std::move(node));
            if (!selfClose) stack.push_back(&stack.back()->children.back());
        }
        if (stack.size() != 1) return Fail(err, "unclosed element <" + stack.back()->name + ">");
        return true;
    }

private:
    const std::string& s;
    size_t pos = 0;
    bool Fail(std::string& err, const std::string& m) {
        err = "XML: " + m;
        return false;
    }
    static std::string Trim(std::string t) {
        while (!t.empty() && std::isspace((unsigned char)t.back())) t.pop_back();
        size_t b = 0;
        while (b < t.size() && std::isspace((unsigned char)t[b])) ++b;
        return t.substr(b);
    }
    void SkipSpace() {
        while (pos < s.size() && std::isspace((unsigned char)s[pos])) ++pos;
    }
    std::string ReadName() {
        size_t b = pos;
        while (pos < s.size() && (std::isalnum((unsigned char)s[pos]) || s[pos] == '_' || s[pos] == '-' || s[pos] == ':' || s[pos] == '.'))
            ++pos;
        return s.substr(b, pos - b);
    }
};

// ---- transforms ----
namespace xf {
inline Matrix34 Scale(float x, float y, float z) {
    Matrix34 m;
    m.m[0] = x; m.m[5] = y; m.m[10] = z;
    return m;
}
inline Matrix34 Translate(float x, float y, float z) {
    Matrix34 m;
    m.m[3] = x; m.m[7] = y; m.m[11] = z;
    return m;
}
inline Matrix34 Rotate(float angleDeg, Vec3f axis) {
    Matrix34 m;
    const float len = axis.Length();
    if (len <= 0.0f) return m;
    axis = axis / len;
    const float a = angleDeg * 3.14159265358979323846f / 180.0f, c = std::cos(a), s = std::sin(a), t = 1.0f - c;
    const float x = axis.x, y = axis.y, z = axis.z;
    const float r[9] = {t * x * x + c, t * x * y - s * z, t * x * z + s * y, t * x * y + s * z, t * y * y + c,
                        t * y * z - s * x, t * x * z - s * y, t * y * z + s * x, t * z * z + c};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) m.m[i * 4 + j] = r[i * 3 + j];
    return m;
}
inline Matrix34 Inverse(Matrix34 const& M) {
    const float* m = M.m;
    const double a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], h = m[9], i = m[10];
    const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    const double id = 1.0 / det;
    const double r[9] = {(e * i - f * h) * id, (c * h - b * i) * id, (b * f - c * e) * id, (f * g - d * i) * id, (a * i - c * g) * id,
                         (c * d - a * f) * id, (d * h - e * g) * id, (b * g - a * h) * id, (a * e - b * d) * id};
    Matrix34 R;
    for (int k = 0; k < 3; ++k) {
        for (int j = 0; j < 3; ++j) R.m[k * 4 + j] = (float)r[k * 3 + j];
        R.m[k * 4 + 3] = (float)(-(r[k * 3] * m[3] + r[k * 3 + 1] * m[7] + r[k * 3 + 2] * m[11]));
    }
    return R;
}
// <scale>, <rotate>, <translate> children of e, composed in document order onto tm; returns false for other names
inline bool Apply(const XmlNode& c, Matrix34& tm) {
    if (c.name == "scale") {
        const float v = c.Float("value", 1.0f);
        tm = Matrix34::Compose(Scale(c.Float("x", 1.0f) * v, c.Float("y", 1.0f) * v, c.Float("z", 1.0f) * v), tm);
    } else if (c.name == "rotate") {
        tm = Matrix34::Compose(Rotate(c.Float("angle", 0.0f), Vec3f(c.Float("x", 0.0f), c.Float("y", 0.0f), c.Float("z", 0.0f))), tm);
    } else if (c.name == "translate") {
        tm = Matrix34::Compose(Translate(c.Float("x", 0.0f), c.Float("y", 0.0f), c.Float("z", 0.0f)), tm);
    } else
        return false;
    return true;
}
}  // namespace xf

// ---- Wavefront OBJ: v / vn / vt / f (a/b/c, a//c, a/b, negative indices, polygons as fans) / usemtl ----
inline bool LoadObj(const std::string& path, TriObj& out, std::string& err) {
    std::ifstream in(path);
    if (!in) {
        err = "cannot open " + path;
        return false;
    }
    std::string head;
    std::getline(in, head);
    if (head.rfind("version https://git-lfs", 0) == 0) {
        err = path + " is a git-lfs pointer, not a mesh";
        return false;
    }
    in.seekg(0);
    out = TriObj();
    std::map<std::string, int> mtlIndex;
    int curMtl = 0;
    bool anyMtl = false;
    std::string line;
    while (std::getline(in, line)) {
        const char* p = line.c_str();
        while (*p == ' ' || *p == '\t') ++p;
        if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
            Vec3f v;
            if (std::sscanf(p + 2, "%f %f %f", &v.x, &v.y, &v.z) == 3) out.v.push_back(v);
        } else if (p[0] == 'v' && p[1] == 'n') {
            Vec3f v;
            if (std::sscanf(p + 3, "%f %f %f", &v.x, &v.y, &v.z) == 3) out.vn.push_back(v);
        } else if (p[0] == 'v' && p[1] == 't') {
            Vec3f v;
            const int n = std::sscanf(p + 3, "%f %f %f", &v.x, &v.y, &v.z);
            if (n >= 2) out.vt.push_back(v);
        } else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
            std::vector<int> vi, ti, ni;
            std::istringstream ss(p + 2);
            std::string tok;
            while (ss >> tok) {
                int a = 0, b = 0, c = 0;
                const char* t = tok.c_str();
                a = std::atoi(t);
                const char* s1 = std::strchr(t, '/');
                if (s1) {
                    if (s1[1] != '/') b = std::atoi(s1 + 1);
                    const char* s2 = std::strchr(s1 + 1, '/');
                    if (s2) c = std::atoi(s2 + 1);
                }
                auto fix = [](int i, size_t n) { return i > 0 ? i - 1 : (i < 0 ? (int)n + i : -1); };
                vi.push_back(fix(a, out.v.size()));
                ti.push_back(fix(b, out.vt.size()));
                ni.push_back(fix(c, out.vn.size()));
            }
            for (size_t k = 2; k < vi.size(); ++k) {
                const int ia[3] = {vi[0], vi[k - 1], vi[k]}, it[3] = {ti[0], ti[k - 1], ti[k]}, in_[3] = {ni[0], ni[k - 1], ni[k]};
                if (ia[0] < 0 || ia[1] < 0 || ia[2] < 0 || ia[0] >= (int)out.v.size() || ia[1] >= (int)out.v.size() || ia[2] >= (int)out.v.size()) {
                    err = path + ": face references a missing vertex";
                    return false;
                }
                out.f.push_back({{ia[0], ia[1], ia[2]}});
                out.ft.push_back({{it[0], it[1], it[2]}});
                out.fn.push_back({{in_[0], in_[1], in_[2]}});
                out.faceMtl.push_back(curMtl);
            }
        } else if (std::strncmp(p, "usemtl", 6) == 0) {
            std::string n = line.substr((p - line.c_str()) + 6);
            while (!n.empty() && std::isspace((unsigned char)n.front())) n.erase(n.begin());
            while (!n.empty() && std::isspace((unsigned char)n.back())) n.pop_back();
            auto it = mtlIndex.find(n);
            if (it == mtlIndex.end()) it = mtlIndex.emplace(n, (int)mtlIndex.size()).first;
            curMtl = it->second;
            anyMtl = true;
        }
    }
    // index arrays are all-or-nothing per mesh: drop normals / texcoords when some face lacks them
    auto complete = [](std::vector<TriObj::TriFace> const& f, size_t n) {
        for (auto const& t : f)
            for (int k = 0; k < 3; ++k)
                if (t.v[k] < 0 || t.v[k] >= (int)n) return false;
        return !f.empty();
    };
    if (!complete(out.fn, out.vn.size())) {
        out.fn.clear();
        out.vn.clear();
    }
    if (!complete(out.ft, out.vt.size())) {
        out.ft.clear();
        out.vt.clear();
    }
    if (!anyMtl) out.faceMtl.clear();
    out.BuildAcceleration();
    return true;
}

// ---- textures: PPM (P6, 8 bit) and PFM (PF, little or big endian, bottom-up) ----
inline std::shared_ptr<TextureMap> LoadPnm(const std::string& path) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return nullptr;
    auto tex = std::make_shared<TextureMap>();
    char magic[3] = {0};
    int w = 0, h = 0;
    float scale = 0;
    bool ok = std::fscanf(f, "%2s", magic) == 1;
    auto skipComments = [&] {
        int c;
        while ((c = std::fgetc(f)) != EOF) {
            if (c == '#') {
                while ((c = std::fgetc(f)) != EOF && c != '\n') {}
            } else if (!std::isspace(c)) {
                std::ungetc(c, f);
                break;
            }
        }
    };
    if (ok && std::strcmp(magic, "P6") == 0) {
        int maxv = 0;
        skipComments();
        ok = std::fscanf(f, "%d", &w) == 1;
        skipComments();
        ok = ok && std::fscanf(f, "%d", &h) == 1;
        skipComments();
        ok = ok && std::fscanf(f, "%d", &maxv) == 1 && maxv == 255 && w > 0 && h > 0;
        std::fgetc(f);
        if (ok) {
            tex->rgb8.resize((size_t)w * h * 3);
            ok = std::fread(tex->rgb8.data(), 1, tex->rgb8.size(), f) == tex->rgb8.size();
        }
    } else if (ok && std::strcmp(magic, "PF") == 0) {
        ok = std::fscanf(f, "%d %d %f", &w, &h, &scale) == 3 && w > 0 && h > 0;
        std::fgetc(f);
        if (ok) {
            tex->isFloat = true;
            tex->rgb32f.resize((size_t)w * h * 3);
            std::vector<float> row((size_t)w * 3);
            for (int y = h - 1; y >= 0 && ok; --y) {   // PFM stores the bottom row first
                ok = std::fread(row.data(), 4, row.size(), f) == row.size();
                if (scale > 0)   // big endian
                    for (float& v : row) {
                        unsigned char* b = reinterpret_cast<unsigned char*>(&v);
                        std::swap(b[0], b[3]);
                        std::swap(b[1], b[2]);
                    }
                std::memcpy(&tex->rgb32f[(size_t)y * w * 3], row.data(), row.size() * 4);
            }
        }
    } else
        ok = false;
    std::fclose(f);
    if (!ok) return nullptr;
    tex->width = w;
    tex->height = h;
    return tex;
}

struct LoadedScene {
    Scene scene;
    Camera camera;
    std::map<std::string, std::shared_ptr<Material>> materials;
    std::vector<std::string> warnings;   // what was skipped (missing meshes / textures, unsupported elements)
};

class SceneLoader {
public:
    // returns a texture for a path relative to the scene file, or null (then the value alone is used)
    std::function<std::shared_ptr<TextureMap>(const std::string&)> textureResolver;
    bool allowMissingMeshes = false;   // keep going when an OBJ cannot be read (e.g. git-lfs pointers)

    bool Load(const std::string& xmlPath, LoadedScene& out, std::string& err) {
        std::ifstream in(xmlPath);
        if (!in) {
            err = "cannot open " + xmlPath;
            return false;
        }
        std::stringstream buf;
        buf << in.rdbuf();
        const size_t slash = xmlPath.find_last_of('/');
        baseDir = slash == std::string::npos ? std::string() : xmlPath.substr(0, slash + 1);
        return LoadText(buf.str(), out, err);
    }

    bool LoadText(const std::string& text, LoadedScene& out, std::string& err) {
        XmlNode doc;
        XmlParser parser(text);
        if (!parser.Parse(doc, err)) return false;
        const XmlNode* xml = doc.Child("xml");
        if (!xml) {
            err = "no <xml> element";
            return false;
        }
        const XmlNode* sc = xml->Child("scene");
        if (!sc) {
            err = "no <scene> element";
            return false;
        }
        ls = &out;
        textures.clear();
        meshes.clear();
        // materials first: objects refer to them by name
        for (auto const& e : sc->children)
            if (e.name == "material" && !ReadMaterial(e, err)) return false;
        for (auto const& e : sc->children) {
            if (e.name == "object") {
                if (!ReadObject(e, out.scene.rootNode, err)) return false;
            } else if (e.name == "light")
                ReadLight(e);
            else if (e.name == "environment")
                ReadTexturedColor(e, out.scene.environment, true);
            else if (e.name == "background") {
                ReadTexturedColor(e, out.scene.background, false);
                out.scene.backgroundEmission = e.Float("emission", 0.0f);
            } else if (e.name != "material")
                out.warnings.push_back("ignored <" + e.name + ">");
        }
        if (const XmlNode* cam = xml->Child("camera")) ReadCamera(*cam, out.camera);
        return true;
    }

private:
    std::string baseDir;
    LoadedScene* ls = nullptr;
    std::map<std::string, std::shared_ptr<TextureMap>> textures;
    std::map<std::string, std::shared_ptr<TriObj>> meshes;

    static Color ReadColor(const XmlNode& e, Color def = Color(1, 1, 1)) {
        Color c(e.Float("r", def.r), e.Float("g", def.g), e.Float("b", def.b));
        const float v = e.Float("value", 1.0f);
        return Color(c.r * v, c.g * v, c.b * v);
    }
    std::shared_ptr<TextureMap> GetTexture(const std::string& name) {
        auto it = textures.find(name);
        if (it != textures.end()) return it->second;
        std::shared_ptr<TextureMap> t;
        if (textureResolver) t = textureResolver(baseDir + name);
        if (!t) t = LoadPnm(baseDir + name);
        if (!t) t = imgdec::LoadPng(baseDir + name);
        if (!t) t = imgdec::LoadHdr(baseDir + name);
        if (!t) ls->warnings.push_back("texture not loaded: " + name);
        textures[name] = t;
        return t;
    }
    // colour, optional texture with its own transform; the map stores the inverse (uvw goes to texture space)
    void ReadTexturedColor(const XmlNode& e, TexturedColor& tc, bool /*environment*/) {
        tc.value = ReadColor(e);
        tc.map.reset();
        if (!e.Has("texture")) return;
        std::shared_ptr<TextureMap> base = GetTexture(e.Str("texture"));
        if (!base) return;
        Matrix34 tm;
        bool any = false;
        for (auto const& c : e.children) any = xf::Apply(c, tm) || any;
        if (any) {
            auto inst = std::make_shared<TextureMap>(*base);   // same texels, own transform
            inst->itm = xf::Inverse(tm);
            tc.map = inst;
        } else
            tc.map = base;
    }
    bool ReadMaterial(const XmlNode& e, std::string& err) {
        const std::string type = e.Str("type", "blinn"), name = e.Str("name");
        std::shared_ptr<Material> m;
        if (type == "blinn") m = std::make_shared<MtlBlinn>();
        else if (type == "phong") m = std::make_shared<MtlPhong>();
        else {
            err = "material '" + name + "': unsupported type '" + type + "'";
            return false;
        }
        // framework defaults: no reflection / refraction / emission
        m->reflection.value = m->refraction.value = m->emission.value = Color(0, 0, 0);
        for (auto const& c : e.children) {
            if (c.name == "diffuse") ReadTexturedColor(c, m->diffuse, false);
            else if (c.name == "specular") ReadTexturedColor(c, m->specular, false);
            else if (c.name == "reflection") ReadTexturedColor(c, m->reflection, false);
            else if (c.name == "refraction") {
                ReadTexturedColor(c, m->refraction, false);
                m->ior = c.Float("index", m->ior);
            } else if (c.name == "emission") ReadTexturedColor(c, m->emission, false);
            else if (c.name == "glossiness") {
                m->glossiness = c.Float("value", m->glossiness);
                if (c.Has("texture")) m->glossinessMap = GetTexture(c.Str("texture"));
            }
            else if (c.name == "absorption") m->absorption = ReadColor(c, Color(0, 0, 0));
            else if (c.name == "bump") {
                if (c.Has("texture")) m->bump = GetTexture(c.Str("texture"));
                m->bumpIntensity = c.Float("value", 1.0f);
            } else
                ls->warnings.push_back("material '" + name + "': ignored <" + c.name + ">");
        }
        ls->materials[name] = m;
        return true;
    }
    bool ReadObject(const XmlNode& e, Node& parent, std::string& err) {
        Node* node = new Node();
        parent.AppendChild(node);
        const std::string type = e.Str("type"), name = e.Str("name");
        if (type == "obj") {
            auto it = meshes.find(name);
            if (it == meshes.end()) {
                auto mesh = std::make_shared<TriObj>();
                std::string oerr;
                if (!LoadObj(baseDir + name, *mesh, oerr)) {
                    if (!allowMissingMeshes) {
                        err = oerr;
                        return false;
                    }
                    ls->warnings.push_back("mesh not loaded: " + oerr);
                    mesh.reset();
                }
                it = meshes.emplace(name, mesh).first;
            }
            if (it->second) node->SetNodeObj(it->second);
        } else if (type == "sphere")
            node->SetNodeObj(std::make_shared<Sphere>());
        else if (type == "plane")
            node->SetNodeObj(std::make_shared<Plane>());
        else if (!type.empty())
            ls->warnings.push_back("object '" + name + "': unsupported type '" + type + "'");
        if (e.Has("material")) {
            auto it = ls->materials.find(e.Str("material"));
            if (it != ls->materials.end()) node->SetMaterial(it->second);
            else ls->warnings.push_back("object '" + name + "': unknown material '" + e.Str("material") + "'");
        }
        for (auto const& c : e.children) {
            if (c.name == "object") {
                if (!ReadObject(c, *node, err)) return false;
            } else if (!xf::Apply(c, node->tm))
                ls->warnings.push_back("object '" + name + "': ignored <" + c.name + ">");
        }
        return true;
    }
    void ReadLight(const XmlNode& e) {
        if (e.Str("type") != "point") {
            ls->warnings.push_back("light '" + e.Str("name") + "': type '" + e.Str("type") + "' is not on the path-traced path");
            return;
        }
        PointLight* L = ls->scene.AddPointLight();
        for (auto const& c : e.children) {
            if (c.name == "intensity") L->intensity = ReadColor(c);
            else if (c.name == "position") L->position = Vec3f(c.Float("x", 0), c.Float("y", 0), c.Float("z", 0));
            else if (c.name == "size") L->size = c.Float("value", 0.0f);
            else if (c.name == "attenuation") L->attenuation = c.Float("value", 1.0f);
        }
    }
    static void ReadCamera(const XmlNode& e, Camera& cam) {
        Vec3f pos(0, 0, 0), target(0, 0, -1), up(0, 1, 0);
        bool haveFocal = false;
        for (auto const& c : e.children) {
            if (c.name == "position") pos = Vec3f(c.Float("x", 0), c.Float("y", 0), c.Float("z", 0));
            else if (c.name == "target") target = Vec3f(c.Float("x", 0), c.Float("y", 0), c.Float("z", 0));
            else if (c.name == "up") up = Vec3f(c.Float("x", 0), c.Float("y", 0), c.Float("z", 0));
            else if (c.name == "fov") cam.fov = c.Float("value", cam.fov);
            else if (c.name == "focaldist") {
                cam.focaldist = c.Float("value", cam.focaldist);
                haveFocal = true;
            } else if (c.name == "dof") cam.dof = c.Float("value", cam.dof);
            else if (c.name == "width") cam.imgWidth = (int)c.Float("value", (float)cam.imgWidth);
            else if (c.name == "height") cam.imgHeight = (int)c.Float("value", (float)cam.imgHeight);
        }
        cam.upHint = up;
        cam.LookAt(pos, target, haveFocal);
        cam.sRGB = e.Str("gamma") == "sRGB";
        // <keyframe time="..."> position / target / fov </keyframe> (cerealBar.xml:220-271); missing fields keep the
        // static camera's values
        for (auto const& c : e.children) {
            if (c.name != "keyframe") continue;
            Camera::Keyframe k{(double)c.Float("time", 0.0f), pos, target, cam.fov};
            for (auto const& f : c.children) {
                if (f.name == "position") k.pos = Vec3f(f.Float("x", 0), f.Float("y", 0), f.Float("z", 0));
                else if (f.name == "target") k.target = Vec3f(f.Float("x", 0), f.Float("y", 0), f.Float("z", 0));
                else if (f.name == "fov") k.fov = f.Float("value", k.fov);
            }
            cam.keyframes.push_back(k);
        }
        std::stable_sort(cam.keyframes.begin(), cam.keyframes.end(),
                         [](Camera::Keyframe const& a, Camera::Keyframe const& b) { return a.time < b.time; });
    }
};

}  // namespace rtamd
