"""ctypes loader of the CPU oracle (oracle/liboracle.so). TEST INFRASTRUCTURE: imported by tests/, smoke() and
bench.py's cpu_baseline leg only."""
import ctypes as C
import os
import subprocess
import numpy as np

from raytracer_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# ORACLE_LIB selects another build of the oracle (A/B runs of experimental variants)
LIB = os.environ.get("ORACLE_LIB") or os.path.join(ROOT, "oracle", "liboracle.so")
# the leg whose elementary functions are the platform libm (oracle/oracle_math.h, ORC_PLATFORM_LIBM)
LIB_LIBM = os.path.join(ROOT, "oracle", "liboracle_libm.so")
_lib = None
_lib_libm = None
_vp = C.c_void_p


def lib(libm=False):
    """The oracle library; libm=True: the build that calls the platform libm instead of the routines shared with the kernels."""
    global _lib, _lib_libm
    if libm:
        if _lib_libm is None:
            _lib_libm = _load(LIB_LIBM)
        return _lib_libm
    if _lib is None:
        _lib = _load(LIB)
    return _lib


def _load(path):
    if True:
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        L = C.CDLL(path)
        L.orc_halton.restype = C.c_float
        L.orc_halton.argtypes = [C.c_int, C.c_int]
        L.orc_rnd.restype = C.c_float
        L.orc_rnd.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_filter1d.restype = C.c_float
        L.orc_filter1d.argtypes = [C.c_int, C.c_float, C.c_float]
        L.orc_photon_download.restype = C.c_int64
        L.orc_photon_download.argtypes = [_vp, C.c_int, _vp, C.c_size_t]
        L.orc_scene_destroy.argtypes = [_vp]
        L.orc_photon_destroy.argtypes = [_vp]
    return L


class OracleScene:
    def __init__(self, data, libm=False):
        self.L = lib(libm)
        self.data = data
        desc, self._keep = data.to_desc()
        h = _vp()
        self.L.orc_scene_create(C.byref(desc), C.byref(h))
        self.h = h
        self.face_base = data.face_base()

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_scene_destroy(self.h)
            self.h = None

    def set_tree(self, parent, tms, mesh):
        """The scene graph below rootNode (orc_scene_set_tree): from now on intersections walk it level by level."""
        parent = np.ascontiguousarray(parent, np.int32)
        tms = np.ascontiguousarray(tms, np.float32).reshape(-1, 12)
        mesh = np.ascontiguousarray(mesh, np.int32)
        rc = self.L.orc_scene_set_tree(self.h, int(parent.shape[0]), _vp(parent.ctypes.data), _vp(tms.ctypes.data), _vp(mesh.ctypes.data))
        assert rc == 0, "bad node tree"

    def bounds(self):
        b = (C.c_float * 6)()
        self.L.orc_scene_bounds(self.h, b)
        return [float(x) for x in b]

    def trace_closest(self, rays, hit_side=_ffi.HIT_FRONT, with_lights=False, use_bvh=True, counters=False):
        ray_dt, hit_dt, _ = _ffi.np_dtypes()
        rays = np.ascontiguousarray(rays, dtype=ray_dt)
        hits = np.zeros(rays.shape[0], dtype=hit_dt)
        cnt = (C.c_uint64 * 2)()
        self.L.orc_trace_closest(self.h, _vp(rays.ctypes.data), C.c_size_t(rays.shape[0]), hit_side, int(with_lights),
                                 int(use_bvh), _vp(hits.ctypes.data), cnt if counters else None)
        return (hits, (int(cnt[0]), int(cnt[1]))) if counters else hits

    def trace_visibility(self, rays, normals, use_transmittance=True):
        ray_dt, _, _ = _ffi.np_dtypes()
        rays = np.ascontiguousarray(rays, dtype=ray_dt)
        normals = np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
        vis = np.zeros(rays.shape[0], dtype=np.float32)
        self.L.orc_trace_visibility(self.h, _vp(rays.ctypes.data), _vp(normals.ctypes.data), C.c_size_t(rays.shape[0]),
                                    int(use_transmittance), _vp(vis.ctypes.data))
        return vis

    def trace_secondary(self, at, dirs, recursive=True, child=2, seed=0x9E3779B97F4A7C15, photons=None):
        dirs = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 3)
        out = np.zeros((dirs.shape[0], 4), np.float32)
        self.L.orc_trace_secondary(self.h, photons.h if photons else None, C.byref(at), _vp(dirs.ctypes.data),
                                   C.c_size_t(dirs.shape[0]), int(recursive), child, C.c_uint64(seed), _vp(out.ctypes.data))
        return out

    def render_primary(self, cam, hit_side=_ffi.HIT_FRONT, use_bvh=True, counters=False):
        n = cam.width * cam.height
        prim = np.zeros(n, dtype=np.int32)
        t = np.zeros(n, dtype=np.float32)
        front = np.zeros(n, dtype=np.uint8)
        cnt = (C.c_uint64 * 2)()
        self.L.orc_render_primary(self.h, C.byref(cam), hit_side, int(use_bvh), _vp(prim.ctypes.data), _vp(t.ctypes.data),
                                  _vp(front.ctypes.data), cnt if counters else None)
        shape = (cam.height, cam.width)
        res = (prim.reshape(shape), t.reshape(shape), front.reshape(shape))
        return res + ((int(cnt[0]), int(cnt[1])),) if counters else res

    def render(self, cam, params, photons=None, nthreads=1, subset_stride=1):
        n = cam.width * cam.height
        accum = np.zeros((n, 4), np.float32)
        count = np.zeros(n, np.int32)
        z = np.zeros(n, np.float32)
        albedo = np.zeros((n, 4), np.float32)
        normal = np.zeros((n, 4), np.float32)
        linear = np.zeros((n, 3), np.float32)
        rgb8 = np.zeros((n, 3), np.uint8)
        cnt = (C.c_uint64 * 3)()
        secs = C.c_double()
        self.L.orc_render(self.h, C.byref(cam), C.byref(params), photons.h if photons else None, nthreads, subset_stride,
                          _vp(accum.ctypes.data), _vp(count.ctypes.data), _vp(z.ctypes.data), _vp(albedo.ctypes.data),
                          _vp(normal.ctypes.data), _vp(linear.ctypes.data), _vp(rgb8.ctypes.data), cnt, C.byref(secs))
        shape = (cam.height, cam.width)
        return dict(accum=accum.reshape(shape + (4,)), samples=count.reshape(shape), z=z.reshape(shape),
                    albedo=albedo.reshape(shape + (4,)), normal=normal.reshape(shape + (4,)),
                    linear=linear.reshape(shape + (3,)), rgb8=rgb8.reshape(shape + (3,)),
                    rays=dict(primary=int(cnt[0]), bounce=int(cnt[1]), shadow=int(cnt[2])), seconds=secs.value)

    def trace_samples(self, cam, params, xys, photons=None):
        xys = np.ascontiguousarray(xys, dtype=np.int32).reshape(-1, 3)
        out = np.zeros((xys.shape[0], 4), np.float32)
        self.L.orc_trace_samples(self.h, C.byref(cam), C.byref(params), photons.h if photons else None,
                                 _vp(xys.ctypes.data), C.c_size_t(xys.shape[0]), _vp(out.ctypes.data))
        return out

    def render_samples(self, cam, params, photons=None):
        H, W, spp = cam.height, cam.width, params.spp_min
        ys, xs, ss = np.meshgrid(np.arange(H), np.arange(W), np.arange(spp), indexing="ij")
        xys = np.stack([xs.ravel(), ys.ravel(), ss.ravel()], 1)
        return self.trace_samples(cam, params, xys, photons).reshape(H, W, spp, 4)


class OraclePhotons:
    def __init__(self, scene, num_photons, max_bounces=2, photon_radius=0.0, caustic_radius=0.0, search_count=100,
                 seed=0x9E3779B97F4A7C15, first_photon=0, photon_stride=1, finalize=True):
        self.L = lib()
        p = _ffi.PhotonParams(num_photons, max_bounces, photon_radius, caustic_radius, search_count, seed, first_photon,
                              photon_stride)
        h = _vp()
        self.L.orc_photon_build(scene.h, C.byref(p), C.byref(h))
        self.h = h
        self.num_photons = num_photons
        if finalize:
            self.finalize()

    def finalize(self, total=None):
        self.L.orc_photon_finalize(self.h, int(total or self.num_photons))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_photon_destroy(self.h)
            self.h = None

    def info(self):
        counts = (C.c_uint64 * 3)()
        radii = (C.c_float * 2)()
        self.L.orc_photon_info(self.h, counts, radii)
        return dict(n_global=int(counts[0]), n_caustic=int(counts[1]), rays=int(counts[2]),
                    photon_radius=float(radii[0]), caustic_radius=float(radii[1]))

    def download(self, which):
        _, _, ph_dt = _ffi.np_dtypes()
        n = self.info()["n_caustic" if which else "n_global"]
        out = np.zeros(n, dtype=ph_dt)
        self.L.orc_photon_download(self.h, which, out.ctypes.data, n)
        return out

    def estimate(self, which, positions, normals):
        positions = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
        normals = np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
        out = np.zeros_like(positions)
        self.L.orc_photon_estimate(self.h, which, _vp(positions.ctypes.data), _vp(normals.ctypes.data),
                                   C.c_size_t(positions.shape[0]), _vp(out.ctypes.data))
        return out


def median_denoise(linear, srgb=False):
    """Oracle restatement of Denoiser::ApplyMedian (Denoiser.cpp:19-134)."""
    src = np.ascontiguousarray(linear, dtype=np.float32)
    H, W = src.shape[:2]
    lin = np.empty_like(src)
    rgb8 = np.empty((H, W, 3), dtype=np.uint8)
    lib().orc_median_denoise(_vp(src.ctypes.data), W, H, int(bool(srgb)), _vp(lin.ctypes.data), _vp(rgb8.ctypes.data))
    return lin, rgb8
