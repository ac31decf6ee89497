"""ctypes mirror of include/rtx.h and the loader of librtx_amd.so.

The library is the product's only compute path: there is no Python/NumPy fallback. Loading fails loudly when the
shared object is missing (build it with `python -c "import __graft_entry__ as g; g.build()"` or
`make -C raytracer_amd/csrc`).
"""
import ctypes as C
import os

c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int)

HIT_FRONT, HIT_BACK, HIT_FRONT_AND_BACK = 1, 2, 3
BIGFLOAT = 1.0e30
MTL_BLINN, MTL_PHONG, MTL_MICROFACET = 0, 1, 2
OBJ_MESH, OBJ_SPHERE, OBJ_PLANE = 0, 1, 2
TEX_RGB8, TEX_RGB32F = 0, 1
FILTER_TENT, FILTER_GAUSSIAN, FILTER_MITCHELL, FILTER_LANCZOS = 0, 1, 2, 3


class MeshDesc(C.Structure):
    _fields_ = [("v", c_float_p), ("nv", C.c_int), ("vn", c_float_p), ("nvn", C.c_int), ("vt", c_float_p),
                ("nvt", C.c_int), ("f", c_int_p), ("fn", c_int_p), ("ft", c_int_p), ("face_mtl", c_int_p),
                ("nf", C.c_int), ("xform", C.c_float * 12), ("material", C.c_int), ("kind", C.c_int)]


class TexMapDesc(C.Structure):
    _fields_ = [("format", C.c_int), ("width", C.c_int), ("height", C.c_int), ("texels", C.c_void_p),
                ("xform", C.c_float * 12)]


class MaterialDesc(C.Structure):
    _fields_ = [("type", C.c_int),
                ("diffuse", C.c_float * 3), ("diffuse_tex", C.c_int),
                ("specular", C.c_float * 3), ("specular_tex", C.c_int),
                ("reflection", C.c_float * 3), ("reflection_tex", C.c_int),
                ("refraction", C.c_float * 3), ("refraction_tex", C.c_int),
                ("emission", C.c_float * 3), ("emission_tex", C.c_int),
                ("glossiness", C.c_float), ("glossiness_tex", C.c_int), ("ior", C.c_float), ("absorption", C.c_float * 3),
                ("bump_tex", C.c_int), ("bump_scale", C.c_float)]


class LightDesc(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("intensity", C.c_float * 3), ("size", C.c_float),
                ("attenuation", C.c_float)]


class SceneDesc(C.Structure):
    _fields_ = [("meshes", C.POINTER(MeshDesc)), ("n_meshes", C.c_int),
                ("materials", C.POINTER(MaterialDesc)), ("n_materials", C.c_int),
                ("texmaps", C.POINTER(TexMapDesc)), ("n_texmaps", C.c_int),
                ("lights", C.POINTER(LightDesc)), ("n_lights", C.c_int),
                ("environment", C.c_float * 3), ("environment_tex", C.c_int),
                ("background", C.c_float * 3), ("background_tex", C.c_int),
                ("background_emission", C.c_float)]


class Camera(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("dir", C.c_float * 3), ("up", C.c_float * 3), ("fov", C.c_float),
                ("focaldist", C.c_float), ("dof", C.c_float), ("width", C.c_int), ("height", C.c_int),
                ("srgb", C.c_int)]


class Ray(C.Structure):
    _fields_ = [("o", C.c_float * 3), ("tmax", C.c_float), ("d", C.c_float * 3), ("pad", C.c_float)]


class Hit(C.Structure):
    _fields_ = [("t", C.c_float), ("obj", C.c_int), ("face", C.c_int), ("mtl_id", C.c_int),
                ("p", C.c_float * 3), ("N", C.c_float * 3), ("GN", C.c_float * 3), ("uvw", C.c_float * 3),
                ("dpdu", C.c_float * 3), ("dpdv", C.c_float * 3), ("bu", C.c_float), ("bv", C.c_float),
                ("front", C.c_int), ("light", C.c_int)]


SHADING_PATH, SHADING_LEGACY = 0, 1


class RenderParams(C.Structure):
    _fields_ = [("spp_min", C.c_int), ("spp_max", C.c_int), ("index_stride", C.c_int), ("ci_halfwidth", C.c_float),
                ("filter_type", C.c_int), ("filter_radius", C.c_float), ("max_bounces", C.c_int),
                ("seed", C.c_uint64), ("tile_size", C.c_int), ("rank", C.c_int), ("world", C.c_int),
                ("samples_per_pass", C.c_int), ("use_photons", C.c_int), ("shading_mode", C.c_int)]


class ShadePoint(C.Structure):
    _fields_ = [("P", C.c_float * 3), ("N", C.c_float * 3), ("obj", C.c_int), ("bounce", C.c_int), ("diffuse_depth", C.c_int),
                ("pixel", C.c_int), ("sample", C.c_int), ("node", C.c_uint)]


class Film(C.Structure):
    _fields_ = [("accum", C.c_void_p), ("count", C.c_void_p), ("z", C.c_void_p), ("albedo", C.c_void_p),
                ("normal", C.c_void_p)]


class PixelStats(C.Structure):
    _fields_ = [("sum", C.c_void_p), ("sum2", C.c_void_p)]


class Stats(C.Structure):
    _fields_ = [("rays_primary", C.c_uint64), ("rays_bounce", C.c_uint64), ("rays_shadow", C.c_uint64),
                ("rays_photon", C.c_uint64), ("paths", C.c_uint64), ("nodes_visited", C.c_uint64),
                ("tris_tested", C.c_uint64), ("extend_ms", C.c_double), ("shadow_ms", C.c_double),
                ("shade_ms", C.c_double), ("total_ms", C.c_double), ("extend_launches", C.c_uint64),
                ("rays_extend", C.c_uint64), ("shade_launches", C.c_uint64), ("vertices", C.c_uint64),
                ("shadow_launches", C.c_uint64), ("shadow_segments", C.c_uint64), ("shadow_nodes", C.c_uint64),
                ("shadow_tris", C.c_uint64), ("nee_ms", C.c_double), ("repacks", C.c_uint64)]


class PhotonParams(C.Structure):
    _fields_ = [("num_photons", C.c_int), ("max_bounces", C.c_int), ("photon_radius", C.c_float),
                ("caustic_radius", C.c_float), ("search_count", C.c_int), ("seed", C.c_uint64),
                ("first_photon", C.c_int), ("photon_stride", C.c_int)]


class Photon(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("dir", C.c_float * 3), ("power", C.c_float * 3)]


# numpy dtypes matching Ray / Hit / Photon (for zero-copy array views)
def np_dtypes():
    import numpy as np
    ray = np.dtype([("o", "<f4", 3), ("tmax", "<f4"), ("d", "<f4", 3), ("pad", "<f4")])
    hit = np.dtype([("t", "<f4"), ("obj", "<i4"), ("face", "<i4"), ("mtl_id", "<i4"), ("p", "<f4", 3),
                    ("N", "<f4", 3), ("GN", "<f4", 3), ("uvw", "<f4", 3), ("dpdu", "<f4", 3), ("dpdv", "<f4", 3),
                    ("bu", "<f4"), ("bv", "<f4"), ("front", "<i4"), ("light", "<i4")])
    photon = np.dtype([("pos", "<f4", 3), ("dir", "<f4", 3), ("power", "<f4", 3)])
    assert ray.itemsize == C.sizeof(Ray) and hit.itemsize == C.sizeof(Hit) and photon.itemsize == C.sizeof(Photon)
    return ray, hit, photon


_HERE = os.path.dirname(os.path.abspath(__file__))
# RTX_AMD_LIB selects another build of the same library (A/B runs of kernel variants, tools/build_variant.sh)
LIB_PATH = os.environ.get("RTX_AMD_LIB") or os.path.join(_HERE, "librtx_amd.so")

# every symbol include/rtx.h declares: (name, restype, argtypes)
_vp = C.c_void_p
SYMBOLS = [
    ("rtx_device_count", C.c_int, []),
    ("rtx_set_device", C.c_int, [C.c_int]),
    ("rtx_last_error", C.c_char_p, []),
    ("rtx_version", C.c_char_p, []),
    ("rtx_scene_create", C.c_int, [C.POINTER(SceneDesc), C.POINTER(_vp)]),
    ("rtx_scene_destroy", None, [_vp]),
    ("rtx_scene_info", C.c_int, [_vp, C.POINTER(C.c_uint64), c_float_p]),
    ("rtx_trace_closest", C.c_int, [_vp, _vp, C.c_size_t, C.c_int, C.c_int, _vp]),
    ("rtx_trace_closest_device", C.c_int, [_vp, _vp, C.c_size_t, C.c_int, C.c_int, _vp, _vp]),
    ("rtx_trace_visibility", C.c_int, [_vp, _vp, _vp, C.c_size_t, C.c_int, _vp]),
    ("rtx_trace_secondary", C.c_int, [_vp, _vp, _vp, _vp, C.c_size_t, C.c_int, C.c_int, C.c_uint64, _vp]),
    ("rtx_render_primary", C.c_int, [_vp, C.POINTER(Camera), C.c_int, _vp, _vp, _vp]),
    ("rtx_film_clear", C.c_int, [C.POINTER(Film), C.c_int, C.c_int, _vp]),
    ("rtx_render_accumulate", C.c_int, [_vp, C.POINTER(Camera), C.POINTER(RenderParams), _vp, C.POINTER(Film),
                                        C.POINTER(Stats)]),
    ("rtx_film_resolve", C.c_int, [C.POINTER(Film), C.POINTER(Camera), _vp, _vp, _vp, _vp, _vp, _vp]),
    ("rtx_median_denoise", C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    ("rtx_film_create", C.c_int, [C.c_int, C.c_int, C.POINTER(Film), C.POINTER(PixelStats)]),
    ("rtx_film_destroy", None, [C.POINTER(Film), C.POINTER(PixelStats)]),
    ("rtx_render_round", C.c_int, [_vp, C.POINTER(Camera), C.POINTER(RenderParams), _vp, C.POINTER(Film), C.POINTER(PixelStats),
                                   c_int_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.POINTER(Stats)]),
    ("rtx_refine_list", C.c_int64, [C.POINTER(PixelStats), C.c_int, C.c_int, C.c_float, C.c_int, c_int_p, C.c_size_t]),
    ("rtx_refine_list_noalloc", C.c_int, [C.POINTER(PixelStats), C.c_int, C.c_int, C.c_float, C.c_int, _vp, C.c_size_t, _vp, _vp]),
    ("rtx_device_synchronize", C.c_int, []),
    ("rtx_stream_create", C.c_int, [C.POINTER(_vp)]),
    ("rtx_stream_synchronize", C.c_int, [_vp]),
    ("rtx_stream_destroy", C.c_int, [_vp]),
    ("rtx_cancel", C.c_int, [_vp]),
    ("rtx_render", C.c_int, [_vp, C.POINTER(Camera), C.POINTER(RenderParams), _vp, _vp, _vp, _vp, _vp,
                             C.POINTER(Stats)]),
    ("rtx_render_samples", C.c_int, [_vp, C.POINTER(Camera), C.POINTER(RenderParams), _vp, _vp, C.POINTER(Stats)]),
    ("rtx_photon_build", C.c_int, [_vp, C.POINTER(PhotonParams), C.POINTER(_vp), C.POINTER(Stats)]),
    ("rtx_photons_destroy", None, [_vp]),
    ("rtx_photon_info", C.c_int, [_vp, C.POINTER(C.c_uint64), c_float_p]),
    ("rtx_photon_download", C.c_int64, [_vp, C.c_int, _vp, C.c_size_t]),
    ("rtx_photon_copy_raw", C.c_int64, [_vp, C.c_int, _vp, C.c_size_t]),
    ("rtx_photon_set", C.c_int, [_vp, C.c_int, _vp, C.c_size_t, C.c_int]),
    ("rtx_photon_device_ptr", _vp, [_vp, C.c_int]),
    ("rtx_photon_estimate", C.c_int, [_vp, C.c_int, _vp, _vp, C.c_size_t, _vp]),
    ("rtx_group_create", C.c_int, [C.POINTER(SceneDesc), c_int_p, C.c_int, C.POINTER(_vp)]),
    ("rtx_group_destroy", None, [_vp]),
    ("rtx_group_size", C.c_int, [_vp]),
    ("rtx_group_scene", _vp, [_vp, C.c_int]),
    ("rtx_group_photons", _vp, [_vp, C.c_int]),
    ("rtx_group_render", C.c_int, [_vp, C.POINTER(Camera), C.POINTER(RenderParams), _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(Stats)]),
    ("rtx_group_photon_build", C.c_int, [_vp, C.POINTER(PhotonParams), C.POINTER(Stats)]),
    ("rtx_group_cancel", C.c_int, [_vp]),
    ("rtx_set_counters", C.c_int, [_vp, C.c_int]),
    ("rtx_get_counters", C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    ("rtx_debug_validate_bvh", C.c_int, [C.POINTER(SceneDesc), C.POINTER(C.c_uint64)]),
    ("rtx_debug_host_walk", C.c_int, [C.POINTER(SceneDesc), _vp, C.c_size_t, _vp, _vp, C.POINTER(C.c_uint64)]),
    ("rtx_debug_math", C.c_int, [C.c_int, _vp, _vp, C.c_size_t, _vp, C.c_int]),
    ("rtx_debug_queue_layout", C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ("rtx_debug_queue_layout_run", C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ("rtx_debug_pixel_list", C.c_int64, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_size_t]),
    ("rtx_debug_lpt_assign", C.c_int, [_vp, C.c_size_t, C.c_int, _vp]),
]

_lib = None


def load_library(path=None):
    """Loads librtx_amd.so and binds every symbol of include/rtx.h. Raises if the library or a symbol is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    # PyTorch wheels bundle their own HIP runtime (torch/lib/libamdhip64.so, no soname version) next to the system one
    # this library links (libamdhip64.so.7). Both can live in one process, but only when torch's libraries are mapped
    # before the system runtime initialises the GPU; the other order leaves torch with "No HIP GPUs are available".
    # Python callers of this package use torch for device buffers, so map it first when it is installed.
    if not os.environ.get("RTX_AMD_NO_TORCH_PRELOAD"):
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    if not os.path.exists(p):
        raise RuntimeError(
            f"{p} not found: the HIP library is the only compute path of raytracer_amd; build it with "
            "`make -C raytracer_amd/csrc` (or __graft_entry__.build())")
    lib = C.CDLL(p)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib
