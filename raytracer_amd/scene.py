"""Host-side scene description: the flattened (Node, TriObj) list that CollectSceneNodeObjects produces
(reference RayTracer/SceneAdapter.h:8-18) plus materials, lights, environment and camera, as plain arrays.

`SceneData.to_desc()` builds the `rtx_scene_desc` POD that crosses the C ABI (include/rtx.h).
"""
import ctypes as C
import numpy as np

from . import _ffi

IDENTITY = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], dtype=np.float32)


class Mesh:
    def __init__(self, v, f, vn=None, vt=None, fn=None, ft=None, face_mtl=None, xform=None, material=-1, name="",
                 kind=_ffi.OBJ_MESH):
        self.kind = int(kind)
        self.v = np.ascontiguousarray(v, dtype=np.float32).reshape(-1, 3)
        self.f = np.ascontiguousarray(f, dtype=np.int32).reshape(-1, 3)
        self.vn = None if vn is None else np.ascontiguousarray(vn, dtype=np.float32).reshape(-1, 3)
        self.vt = None if vt is None else np.ascontiguousarray(vt, dtype=np.float32).reshape(-1, 3)
        self.fn = None if fn is None else np.ascontiguousarray(fn, dtype=np.int32).reshape(-1, 3)
        self.ft = None if ft is None else np.ascontiguousarray(ft, dtype=np.int32).reshape(-1, 3)
        self.face_mtl = None if face_mtl is None else np.ascontiguousarray(face_mtl, dtype=np.int32)
        self.xform = IDENTITY.copy() if xform is None else np.ascontiguousarray(xform, dtype=np.float32).reshape(12)
        self.material = int(material)
        self.name = name

    @property
    def nf(self):
        return self.f.shape[0] if self.kind == _ffi.OBJ_MESH else 1   # an analytic object is one primitive

    @classmethod
    def sphere(cls, xform=None, material=-1, name="sphere"):
        """Unit sphere in object space (Sphere, objects.cpp:21-86) under the object->world transform `xform`."""
        return cls(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32), xform=xform, material=material, name=name,
                   kind=_ffi.OBJ_SPHERE)

    @classmethod
    def plane(cls, xform=None, material=-1, name="plane"):
        """Square [-1,1]^2 at z = 0 in object space (Plane, objects.cpp:89-126)."""
        return cls(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32), xform=xform, material=material, name=name,
                   kind=_ffi.OBJ_PLANE)


class Material:
    def __init__(self, name="", type=_ffi.MTL_BLINN, diffuse=(0.5, 0.5, 0.5), specular=(0.7, 0.7, 0.7),
                 reflection=(0, 0, 0), refraction=(0, 0, 0), emission=(0, 0, 0), glossiness=20.0, ior=1.5,
                 absorption=(0, 0, 0), diffuse_tex=-1, specular_tex=-1, reflection_tex=-1, refraction_tex=-1,
                 emission_tex=-1, bump_tex=-1, bump_scale=1.0, glossiness_tex=-1):
        self.__dict__.update(locals())
        del self.__dict__["self"]


class TexMap:
    def __init__(self, texels, xform=None):
        t = np.ascontiguousarray(texels)
        assert t.ndim == 3 and t.shape[2] == 3
        if t.dtype == np.uint8:
            self.format = _ffi.TEX_RGB8
        else:
            t = np.ascontiguousarray(t, dtype=np.float32)
            self.format = _ffi.TEX_RGB32F
        self.texels = t
        self.xform = IDENTITY.copy() if xform is None else np.ascontiguousarray(xform, dtype=np.float32).reshape(12)


class Light:
    def __init__(self, position, intensity, size=0.0, attenuation=1.0):
        self.position = tuple(float(x) for x in position)
        self.intensity = tuple(float(x) for x in intensity)
        self.size = float(size)
        self.attenuation = float(attenuation)


class SceneData:
    def __init__(self):
        self.meshes = []
        self.materials = []
        self.texmaps = []
        self.lights = []
        self.environment = (0.0, 0.0, 0.0)
        self.environment_tex = -1
        self.background = (0.0, 0.0, 0.0)
        self.background_tex = -1
        self.background_emission = 0.0

    @property
    def n_triangles(self):
        return sum(m.nf for m in self.meshes)

    def face_base(self):
        return np.cumsum([0] + [m.nf for m in self.meshes])[:-1]

    def to_desc(self):
        """Returns (rtx_scene_desc, keepalive). The desc borrows the numpy buffers: keep `keepalive` referenced."""
        keep = []
        fp = lambda a: a.ctypes.data_as(_ffi.c_float_p)
        ip = lambda a: a.ctypes.data_as(_ffi.c_int_p)
        meshes = (_ffi.MeshDesc * max(1, len(self.meshes)))()
        for i, m in enumerate(self.meshes):
            d = meshes[i]
            d.v, d.nv = fp(m.v), m.v.shape[0]
            d.f, d.nf = ip(m.f), m.f.shape[0]
            if m.vn is not None:
                d.vn, d.nvn = fp(m.vn), m.vn.shape[0]
                if m.fn is not None:
                    d.fn = ip(m.fn)
            if m.vt is not None:
                d.vt, d.nvt = fp(m.vt), m.vt.shape[0]
                if m.ft is not None:
                    d.ft = ip(m.ft)
            if m.face_mtl is not None:
                d.face_mtl = ip(m.face_mtl)
            d.xform = (C.c_float * 12)(*m.xform.tolist())
            d.material = m.material
            d.kind = m.kind
        mats = (_ffi.MaterialDesc * max(1, len(self.materials)))()
        for i, m in enumerate(self.materials):
            d = mats[i]
            d.type = m.type
            for k in ("diffuse", "specular", "reflection", "refraction", "emission", "absorption"):
                setattr(d, k, (C.c_float * 3)(*[float(x) for x in getattr(m, k)]))
            for k in ("diffuse_tex", "specular_tex", "reflection_tex", "refraction_tex", "emission_tex", "bump_tex", "glossiness_tex"):
                setattr(d, k, int(getattr(m, k)))
            d.glossiness, d.ior, d.bump_scale = float(m.glossiness), float(m.ior), float(m.bump_scale)
        texs = (_ffi.TexMapDesc * max(1, len(self.texmaps)))()
        for i, t in enumerate(self.texmaps):
            d = texs[i]
            d.format, d.height, d.width = t.format, t.texels.shape[0], t.texels.shape[1]
            d.texels = t.texels.ctypes.data
            d.xform = (C.c_float * 12)(*t.xform.tolist())
        lights = (_ffi.LightDesc * max(1, len(self.lights)))()
        for i, l in enumerate(self.lights):
            lights[i].position = (C.c_float * 3)(*l.position)
            lights[i].intensity = (C.c_float * 3)(*l.intensity)
            lights[i].size, lights[i].attenuation = l.size, l.attenuation
        desc = _ffi.SceneDesc()
        desc.meshes, desc.n_meshes = meshes, len(self.meshes)
        desc.materials, desc.n_materials = mats, len(self.materials)
        desc.texmaps, desc.n_texmaps = texs, len(self.texmaps)
        desc.lights, desc.n_lights = lights, len(self.lights)
        desc.environment = (C.c_float * 3)(*[float(x) for x in self.environment])
        desc.environment_tex = self.environment_tex
        desc.background = (C.c_float * 3)(*[float(x) for x in self.background])
        desc.background_tex = self.background_tex
        desc.background_emission = float(self.background_emission)
        keep += [meshes, mats, texs, lights, self]
        return desc, keep


def make_camera(pos, target, up=(0, 1, 0), fov=40.0, focaldist=1.0, dof=0.0, width=200, height=150, srgb=False):
    """Camera as the framework's scene loader leaves it: dir = normalize(target - pos), up re-orthogonalised
    (x = dir ^ up; up = normalize(x ^ dir))."""
    pos = np.asarray(pos, dtype=np.float32)
    d = np.asarray(target, dtype=np.float32) - pos
    d = (d / np.float32(np.sqrt(np.float32(np.dot(d, d))))).astype(np.float32)
    upv = np.asarray(up, dtype=np.float32)
    x = np.cross(d, upv).astype(np.float32)
    u = np.cross(x, d).astype(np.float32)
    u = (u / np.float32(np.sqrt(np.float32(np.dot(u, u))))).astype(np.float32)
    cam = _ffi.Camera()
    cam.pos = (C.c_float * 3)(*pos.tolist())
    cam.dir = (C.c_float * 3)(*d.tolist())
    cam.up = (C.c_float * 3)(*u.tolist())
    cam.fov, cam.focaldist, cam.dof = float(fov), float(focaldist), float(dof)
    cam.width, cam.height, cam.srgb = int(width), int(height), int(bool(srgb))
    return cam


def render_params(spp_min=128, spp_max=None, max_bounces=64, seed=0x9E3779B97F4A7C15, ci_halfwidth=0.005,
                  filter_type=_ffi.FILTER_TENT, filter_radius=1.0, tile_size=16, rank=0, world=1,
                  samples_per_pass=0, index_stride=1024, use_photons=0, shading_mode=_ffi.SHADING_PATH):
    """Defaults are the reference's compile-time constants (RayCaster.cpp:51-56,30-31,358,1152)."""
    p = _ffi.RenderParams()
    p.spp_min = spp_min
    p.spp_max = spp_min if spp_max is None else spp_max
    p.index_stride = index_stride
    p.ci_halfwidth = ci_halfwidth
    p.filter_type, p.filter_radius = filter_type, filter_radius
    p.max_bounces = max_bounces
    p.seed = seed
    p.tile_size, p.rank, p.world = tile_size, rank, world
    p.samples_per_pass = samples_per_pass
    p.use_photons = use_photons
    p.shading_mode = shading_mode   # _ffi.SHADING_LEGACY = the reference's SHADING_USE_LEGACY=1 build
    return p
