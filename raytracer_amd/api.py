"""Thin Python wrapper over the C ABI (include/rtx.h). All compute happens inside librtx_amd.so (HIP, gfx950)."""
import ctypes as C
import numpy as np

from . import _ffi
from .scene import SceneData


class RtxError(RuntimeError):
    pass


def _check(lib, rc):
    if rc != 0:
        raise RtxError(f"rtx error {rc}: {lib.rtx_last_error().decode()}")


def device_count():
    return _ffi.load_library().rtx_device_count()


def set_device(i):
    lib = _ffi.load_library()
    _check(lib, lib.rtx_set_device(i))


class Scene:
    """Device-resident flattened scene (rtx_scene handle)."""

    def __init__(self, data: SceneData):
        self.lib = _ffi.load_library()
        self.data = data
        desc, keep = data.to_desc()
        h = C.c_void_p()
        _check(self.lib, self.lib.rtx_scene_create(C.byref(desc), C.byref(h)))
        self.h = h
        self.face_base = data.face_base()

    def close(self):
        if getattr(self, "h", None):
            self.lib.rtx_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        counts = (C.c_uint64 * 6)()
        bounds = (C.c_float * 6)()
        _check(self.lib, self.lib.rtx_scene_info(self.h, counts, bounds))
        keys = ("triangles", "nodes", "node_bytes", "tri_bytes", "device_bytes", "depth")
        d = dict(zip(keys, [int(c) for c in counts]))
        d["bounds"] = [float(b) for b in bounds]
        return d

    def trace_closest(self, rays, hit_side=_ffi.HIT_FRONT, with_lights=False):
        ray_dt, hit_dt, _ = _ffi.np_dtypes()
        rays = np.ascontiguousarray(rays, dtype=ray_dt)
        hits = np.zeros(rays.shape[0], dtype=hit_dt)
        _check(self.lib, self.lib.rtx_trace_closest(self.h, rays.ctypes.data, rays.shape[0], hit_side, int(with_lights),
                                                    hits.ctypes.data))
        return hits

    def trace_closest_device(self, rays_ptr, n, hits_ptr, hit_side=_ffi.HIT_FRONT_AND_BACK, with_lights=False, stream=None):
        _check(self.lib, self.lib.rtx_trace_closest_device(self.h, rays_ptr, n, hit_side, int(with_lights), hits_ptr, stream))

    def trace_visibility(self, rays, normals, use_transmittance=True):
        ray_dt, _, _ = _ffi.np_dtypes()
        rays = np.ascontiguousarray(rays, dtype=ray_dt)
        normals = np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
        vis = np.zeros(rays.shape[0], dtype=np.float32)
        _check(self.lib, self.lib.rtx_trace_visibility(self.h, rays.ctypes.data, normals.ctypes.data, rays.shape[0],
                                                       int(use_transmittance), vis.ctypes.data))
        return vis

    def trace_secondary(self, at, dirs, recursive=True, child=2, seed=0x9E3779B97F4A7C15, photons=None):
        """ShadeInfoAdapter::TraceSecondaryRay (legacy mode) for directions `dirs` from the shading point `at`
        (_ffi.ShadePoint): (n, 4) array of {r, g, b, dist}."""
        dirs = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 3)
        out = np.zeros((dirs.shape[0], 4), dtype=np.float32)
        _check(self.lib, self.lib.rtx_trace_secondary(self.h, photons.h if photons else None, C.byref(at), dirs.ctypes.data,
                                                      dirs.shape[0], int(recursive), child, seed, out.ctypes.data))
        return out

    def render_primary(self, cam, hit_side=_ffi.HIT_FRONT):
        n = cam.width * cam.height
        prim = np.zeros(n, dtype=np.int32)
        t = np.zeros(n, dtype=np.float32)
        front = np.zeros(n, dtype=np.uint8)
        _check(self.lib, self.lib.rtx_render_primary(self.h, C.byref(cam), hit_side, prim.ctypes.data, t.ctypes.data,
                                                     front.ctypes.data))
        shape = (cam.height, cam.width)
        return prim.reshape(shape), t.reshape(shape), front.reshape(shape)

    def set_counters(self, enabled):
        _check(self.lib, self.lib.rtx_set_counters(self.h, int(enabled)))

    def get_counters(self):
        out = (C.c_uint64 * 3)()
        _check(self.lib, self.lib.rtx_get_counters(self.h, out))
        return {"nodes": int(out[0]), "tris": int(out[1]), "rays": int(out[2])}

    def render(self, cam, params, photons=None):
        """CastRays on one GPU: returns dict(linear, rgb8, z, samples, stats)."""
        n = cam.width * cam.height
        linear = np.zeros((n, 3), dtype=np.float32)
        rgb8 = np.zeros((n, 3), dtype=np.uint8)
        z = np.zeros(n, dtype=np.float32)
        samples = np.zeros(n, dtype=np.int32)
        st = _ffi.Stats()
        _check(self.lib, self.lib.rtx_render(self.h, C.byref(cam), C.byref(params), photons.h if photons else None,
                                             linear.ctypes.data, rgb8.ctypes.data, z.ctypes.data, samples.ctypes.data,
                                             C.byref(st)))
        shape = (cam.height, cam.width)
        return dict(linear=linear.reshape(shape + (3,)), rgb8=rgb8.reshape(shape + (3,)), z=z.reshape(shape),
                    samples=samples.reshape(shape), stats=stats_dict(st))

    def render_accumulate(self, cam, params, film, photons=None):
        st = _ffi.Stats()
        _check(self.lib, self.lib.rtx_render_accumulate(self.h, C.byref(cam), C.byref(params),
                                                        photons.h if photons else None, C.byref(film), C.byref(st)))
        return stats_dict(st)

    def render_round(self, cam, params, film, pstats, pixels, sample_base, n_samples, reset_z=False, photons=None):
        """rtx_render_round: samples [sample_base, sample_base+n_samples) of the listed pixels (int32 array of y*W+x)."""
        pixels = np.ascontiguousarray(pixels, dtype=np.int32)
        st = _ffi.Stats()
        _check(self.lib, self.lib.rtx_render_round(self.h, C.byref(cam), C.byref(params), photons.h if photons else None,
                                                   C.byref(film), C.byref(pstats), pixels.ctypes.data_as(_ffi.c_int_p),
                                                   pixels.shape[0], int(sample_base), int(n_samples), int(bool(reset_z)),
                                                   C.byref(st)))
        return stats_dict(st)

    def render_samples(self, cam, params, photons=None):
        """Per-sample radiance for a fixed spp: array (H, W, spp, 4) = r, g, b, primaryDist."""
        spp = params.spp_min
        out = np.zeros((cam.height, cam.width, spp, 4), dtype=np.float32)
        st = _ffi.Stats()
        _check(self.lib, self.lib.rtx_render_samples(self.h, C.byref(cam), C.byref(params),
                                                     photons.h if photons else None, out.ctypes.data, C.byref(st)))
        return out, stats_dict(st)


class Group:
    """Several GPUs driven from this one process (rtx_group): one scene replica per entry of `devices` (an id may repeat),
    buckets dealt b % n, films merged on the first member's GPU. The C++ host mirror uses the same entry points."""

    def __init__(self, data: SceneData, devices):
        self.lib = _ffi.load_library()
        desc, keep = data.to_desc()
        ids = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        _check(self.lib, self.lib.rtx_group_create(C.byref(desc), ids, len(devices), C.byref(h)))
        self.h = h
        self.devices = list(devices)

    def close(self):
        if getattr(self, "h", None):
            self.lib.rtx_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render(self, cam, params):
        n = cam.width * cam.height
        linear = np.zeros((n, 3), dtype=np.float32)
        rgb8 = np.zeros((n, 3), dtype=np.uint8)
        z = np.zeros(n, dtype=np.float32)
        samples = np.zeros(n, dtype=np.int32)
        albedo = np.zeros((n, 3), dtype=np.float32)
        normal = np.zeros((n, 3), dtype=np.float32)
        st = _ffi.Stats()
        _check(self.lib, self.lib.rtx_group_render(self.h, C.byref(cam), C.byref(params), linear.ctypes.data, rgb8.ctypes.data,
                                                   z.ctypes.data, samples.ctypes.data, albedo.ctypes.data, normal.ctypes.data, C.byref(st)))
        shape = (cam.height, cam.width)
        return dict(linear=linear.reshape(shape + (3,)), rgb8=rgb8.reshape(shape + (3,)), z=z.reshape(shape),
                    samples=samples.reshape(shape), albedo=albedo.reshape(shape + (3,)), normal=normal.reshape(shape + (3,)),
                    stats=stats_dict(st))

    def photon_build(self, num_photons=10_000_000, max_bounces=2, photon_radius=0.0, caustic_radius=0.0, search_count=100,
                     seed=0x9E3779B97F4A7C15):
        p = _ffi.PhotonParams(num_photons, max_bounces, photon_radius, caustic_radius, search_count, seed, 0, 1)
        st = _ffi.Stats()
        _check(self.lib, self.lib.rtx_group_photon_build(self.h, C.byref(p), C.byref(st)))
        return stats_dict(st)

    def member_photons(self, member):
        """Photons view of one member's maps (borrowed: info / estimate / download only)."""
        ph = Photons.__new__(Photons)
        ph.lib = self.lib
        h = self.lib.rtx_group_photons(self.h, member)
        if not h:
            raise RtxError("the group holds no photon map (photon_build first)")
        ph.h = C.c_void_p(h)
        ph.close = lambda: None          # the group owns the handle
        return ph


def stats_dict(st):
    return {k: getattr(st, k) for k, _ in _ffi.Stats._fields_}


def median_denoise(linear, srgb=False):
    """rtx_median_denoise: 3x3 median of a linear RGB image (H, W, 3); returns (linear_out, rgb8_out)."""
    lib = _ffi.load_library()
    src = np.ascontiguousarray(linear, dtype=np.float32)
    H, W = src.shape[:2]
    lin = np.empty_like(src)
    rgb8 = np.empty((H, W, 3), dtype=np.uint8)
    _check(lib, lib.rtx_median_denoise(src.ctypes.data, W, H, int(bool(srgb)), lin.ctypes.data, rgb8.ctypes.data))
    return lin, rgb8


def refine_list(pstats, width, height, ci_halfwidth, spp_max):
    """rtx_refine_list: ascending indices of the pixels whose confidence interval is still too wide."""
    lib = _ffi.load_library()
    out = np.empty(width * height, dtype=np.int32)
    n = lib.rtx_refine_list(C.byref(pstats), width, height, float(ci_halfwidth), int(spp_max),
                            out.ctypes.data_as(_ffi.c_int_p), out.shape[0])
    if n < 0:
        _check(lib, int(n))
    return out[:n].copy()


def refine_list_noalloc(pstats, width, height, ci_halfwidth, spp_max, out_pixels_ptr, capacity, out_count_ptr, stream=None):
    """rtx_refine_list_noalloc: the same test with caller-owned device outputs, enqueued on `stream`; allocates nothing."""
    lib = _ffi.load_library()
    _check(lib, lib.rtx_refine_list_noalloc(C.byref(pstats), width, height, float(ci_halfwidth), int(spp_max), out_pixels_ptr,
                                            capacity, out_count_ptr, stream))


def pixel_list(width, height, tile, rank, world):
    """Pixels of the 16x16 buckets that belong to `rank` (bucket b -> rank b % world), without padding."""
    lib = _ffi.load_library()
    n = lib.rtx_debug_pixel_list(width, height, tile, rank, world, None, 0)
    if n < 0:
        _check(lib, int(n))
    out = np.empty(n, dtype=np.int32)
    lib.rtx_debug_pixel_list(width, height, tile, rank, world, out.ctypes.data, out.shape[0])
    return out[out >= 0]


def film_clear(film, width, height, stream=None):
    lib = _ffi.load_library()
    _check(lib, lib.rtx_film_clear(C.byref(film), width, height, stream))


def film_resolve(film, cam, want_aov=False):
    lib = _ffi.load_library()
    n = cam.width * cam.height
    linear = np.zeros((n, 3), dtype=np.float32)
    rgb8 = np.zeros((n, 3), dtype=np.uint8)
    z = np.zeros(n, dtype=np.float32)
    samples = np.zeros(n, dtype=np.int32)
    albedo = np.zeros((n, 3), dtype=np.float32) if want_aov else None
    normal = np.zeros((n, 3), dtype=np.float32) if want_aov else None
    _check(lib, lib.rtx_film_resolve(C.byref(film), C.byref(cam), linear.ctypes.data, rgb8.ctypes.data, z.ctypes.data,
                                     samples.ctypes.data, albedo.ctypes.data if want_aov else None,
                                     normal.ctypes.data if want_aov else None))
    shape = (cam.height, cam.width)
    out = dict(linear=linear.reshape(shape + (3,)), rgb8=rgb8.reshape(shape + (3,)), z=z.reshape(shape),
               samples=samples.reshape(shape))
    if want_aov:
        out["albedo"] = albedo.reshape(shape + (3,))
        out["normal"] = normal.reshape(shape + (3,))
    return out


class Photons:
    """Photon maps resident in HBM (rtx_photons handle): PhotonMapBuilder::BuildPhotonMap on the GPU."""

    def __init__(self, scene: Scene, num_photons=10_000_000, max_bounces=2, photon_radius=0.0, caustic_radius=0.0,
                 search_count=100, seed=0x9E3779B97F4A7C15, first_photon=0, photon_stride=1):
        self.lib = scene.lib
        self.scene = scene
        p = _ffi.PhotonParams(num_photons, max_bounces, photon_radius, caustic_radius, search_count, seed, first_photon,
                              photon_stride)
        h = C.c_void_p()
        st = _ffi.Stats()
        _check(self.lib, self.lib.rtx_photon_build(scene.h, C.byref(p), C.byref(h), C.byref(st)))
        self.h = h
        self.num_photons = num_photons
        self.stats = stats_dict(st)

    def close(self):
        if getattr(self, "h", None):
            self.lib.rtx_photons_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        counts = (C.c_uint64 * 2)()
        radii = (C.c_float * 2)()
        _check(self.lib, self.lib.rtx_photon_info(self.h, counts, radii))
        return dict(n_global=int(counts[0]), n_caustic=int(counts[1]), photon_radius=float(radii[0]),
                    caustic_radius=float(radii[1]))

    def download(self, which):
        _, _, ph_dt = _ffi.np_dtypes()
        n = self.info()["n_caustic" if which else "n_global"]
        out = np.zeros(n, dtype=ph_dt)
        got = self.lib.rtx_photon_download(self.h, which, out.ctypes.data, n)
        if got < 0:
            raise RtxError(self.lib.rtx_last_error().decode())
        return out[:got]

    def copy_raw(self, which, dst_ptr, capacity):
        """rtx_photon_copy_raw: raw 48-B records of set `which` into a caller-owned device buffer; returns the count."""
        got = self.lib.rtx_photon_copy_raw(self.h, which, dst_ptr, capacity)
        if got < 0:
            raise RtxError(self.lib.rtx_last_error().decode())
        return int(got)

    def set_raw(self, which, src_ptr, n, total_emitted):
        """rtx_photon_set: replace set `which` by n raw records in device memory and rebuild its grid and scaling."""
        _check(self.lib, self.lib.rtx_photon_set(self.h, which, src_ptr, n, int(total_emitted)))

    def estimate(self, which, positions, normals):
        positions = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
        normals = np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
        out = np.zeros_like(positions)
        _check(self.lib, self.lib.rtx_photon_estimate(self.h, which, positions.ctypes.data, normals.ctypes.data,
                                                      positions.shape[0], out.ctypes.data))
        return out
