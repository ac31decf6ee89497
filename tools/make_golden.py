"""Generates tests/golden/*.npz from the CPU oracle on seeded inputs.

The reference holds no tests, golden vectors or fixtures and cannot be compiled here (its hot-path sources include 18
framework headers that are absent from the reference repository), so these fixtures pin the ORACLE against regressions
and against the GPU path; they are not outputs of the reference itself ("parity unpinned")."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from raytracer_amd import synth, _ffi
from raytracer_amd.scene import render_params
from tests.oracle_ffi import OracleScene, OraclePhotons, lib
from tests.helpers import random_rays

out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
os.makedirs(out, exist_ok=True)
L = lib()
rng = np.random.default_rng(20261111)
ray_dt, hit_dt, _ = _ffi.np_dtypes()

def all_fixtures():
    global rng
    # box KAT (scene_box.cpp): random boxes and rays, plus parallel / on-plane / inside / behind cases
    n = 4096
    lo = rng.uniform(-1, 1, (n, 3)).astype(np.float32); hi = (lo + rng.uniform(0, 1, (n, 3))).astype(np.float32)
    rays = np.zeros(n, ray_dt)
    rays["o"] = rng.uniform(-2, 2, (n, 3)); d = rng.normal(size=(n, 3)); rays["d"] = d / np.linalg.norm(d, axis=1, keepdims=True)
    rays["tmax"] = np.where(rng.random(n) < 0.3, rng.uniform(0.1, 3, n), 1e30)
    k = n // 8
    rays["d"][:k, 0] = 0.0                       # parallel to the x slabs
    rays["o"][k:2 * k] = (0.5 * (lo + hi))[k:2 * k]  # origin inside
    rays["o"][2 * k:3 * k, 1] = lo[2 * k:3 * k, 1]   # origin on a slab plane
    rays["d"][2 * k:3 * k, 1] = 0.0
    boxes = np.concatenate([lo, hi], 1).astype(np.float32)
    res = np.zeros(n, np.uint8)
    L.orc_box_intersect(boxes.ctypes.data_as(C.c_void_p), rays.ctypes.data_as(C.c_void_p), C.c_size_t(n), res.ctypes.data_as(C.c_void_p))
    np.savez_compressed(os.path.join(out, "box_kat.npz"), boxes=boxes, rays=rays, hit=res)

    # closest hit on a small generated mesh (C1 style) + brute force + full HitInfo for random rays
    sd, cam = synth.single_mesh_scene("teapots", 0.02)
    cam.width = cam.height = 64
    osc = OracleScene(sd)
    prim, t, front = osc.render_primary(cam, _ffi.HIT_FRONT_AND_BACK)
    rr = random_rays(rng, 2000, osc.bounds())
    hits = osc.trace_closest(rr, _ffi.HIT_FRONT_AND_BACK)
    np.savez_compressed(os.path.join(out, "closest_c1.npz"), prim=prim, t=t, front=front, rays=rr, hits=hits)

    # sampler KAT
    idx = np.array([1, 2, 3, 4, 5, 7, 8, 100, 1023, 1024, 65535, 1 << 20, 2123366400, 2147483647], np.int32)
    hal = np.array([[L.orc_halton(int(i), b) for b in (2, 3, 5, 7)] for i in idx], np.float32)
    px = np.array([[0, 0], [1, 0], [0, 1], [17, 5], [1919, 1079], [640, 360]], np.int32)
    sh = np.zeros((len(px), 2), np.float32)
    for i, (x, y) in enumerate(px):
        a, b = C.c_float(), C.c_float(); L.orc_pixel_shifts(int(x), int(y), C.byref(a), C.byref(b)); sh[i] = (a.value, b.value)
    rnd = np.array([L.orc_rnd(0x9E3779B97F4A7C15, a, b, s, dd) for a, b, s, dd in [(0, 0, 0, 0), (1, 2, 3, 4), (12345, 127, 63, 9), (2**32 - 1, 2**32 - 1, 5, 3)]], np.float32)
    camf = synth.cerealbar_camera()
    srays = np.zeros(len(px) * 3, ray_dt); spos = np.zeros((len(px) * 3, 2), np.float32)
    for i, (x, y) in enumerate(px):
        for s in range(3):
            r = _ffi.Ray(); sp = (C.c_float * 2)()
            L.orc_sample_ray(C.byref(camf), int(x), int(y), s * 63, 1024, C.byref(r), sp)
            srays[i * 3 + s] = (tuple(r.o), r.tmax, tuple(r.d), 0.0); spos[i * 3 + s] = (sp[0], sp[1])
    np.savez_compressed(os.path.join(out, "sampler_kat.npz"), idx=idx, halton=hal, px=px, shifts=sh, rnd=rnd, rays=srays, pos=spos)

    # BSDF KAT on the cerealBar materials + small path-traced render + photons on a tiny cerealbar-synth
    sd = synth.build_scene(0.002)
    osc = OracleScene(sd)
    nm = len(sd.materials)
    ctx = np.zeros((nm * 64, 16), np.float32); so = np.zeros((nm * 64, 10), np.float32); eo = np.zeros((nm * 64, 6), np.float32)
    dirs = np.zeros((nm * 64, 3), np.float32)
    for m in range(nm):
        for j in range(64):
            i = m * 64 + j
            N = rng.normal(size=3); N /= np.linalg.norm(N)
            V = rng.normal(size=3); V /= np.linalg.norm(V)
            if V @ N < 0: V = -V
            front = 1.0 if rng.random() < 0.7 else 0.0
            ctx[i, 0:3] = rng.uniform(-1, 1, 3); ctx[i, 3:6] = V; ctx[i, 6:9] = N if front else -N; ctx[i, 9:12] = ctx[i, 6:9]
            ctx[i, 12:15] = (rng.random(), rng.random(), 0); ctx[i, 15] = front
            L.orc_bsdf_sample(osc.h, m, ctx[i].ctypes.data_as(C.c_void_p), C.c_uint64(0x9E3779B97F4A7C15), i, j, j % 7, so[i].ctypes.data_as(C.c_void_p))
            dd = rng.normal(size=3).astype(np.float32); dirs[i] = dd
            L.orc_bsdf_eval(osc.h, m, ctx[i].ctypes.data_as(C.c_void_p), dirs[i].ctypes.data_as(C.c_void_p), eo[i].ctypes.data_as(C.c_void_p))
    np.savez_compressed(os.path.join(out, "bsdf_kat.npz"), ctx=ctx, sample=so, dirs=dirs, eval=eo)

    cams = synth.cerealbar_camera(48, 27)
    rp = render_params(spp_min=4, max_bounces=64)
    samples = osc.render_samples(cams, rp)
    img = osc.render(cams, rp)
    rpa = render_params(spp_min=8, spp_max=32, max_bounces=8, ci_halfwidth=0.02)
    ada = osc.render(cams, rpa)
    np.savez_compressed(os.path.join(out, "path_small.npz"), samples=samples, linear=img["linear"], z=img["z"], rgb8=img["rgb8"],
                        rays=np.array([img["rays"][k] for k in ("primary", "bounce", "shadow")], np.int64), adaptive_samples=ada["samples"])
    ph = OraclePhotons(osc, 100000)
    g, c = ph.download(0), ph.download(1)
    probe = np.concatenate([g["pos"][:64], c["pos"][:64]]).astype(np.float32)
    nn = np.tile(np.array([[0, 1, 0]], np.float32), (len(probe), 1))
    np.savez_compressed(os.path.join(out, "photon_small.npz"), n_global=len(g), n_caustic=len(c), info_radii=np.array([ph.info()["photon_radius"], ph.info()["caustic_radius"]], np.float32),
                        probe=probe, est_global=ph.estimate(0, probe, nn), est_caustic=ph.estimate(1, probe, nn),
                        sum_power_global=g["power"].astype(np.float64).sum(0), sum_power_caustic=c["power"].astype(np.float64).sum(0))
    print("golden written:", sorted(os.listdir(out)))

# legacy shading mode (SHADING_USE_LEGACY=1): per-sample radiance and ShadeInfoAdapter::TraceSecondaryRay. Run with
# `python tools/make_golden.py legacy` to write only this fixture.
def legacy_fixture():
    sd = synth.build_scene(0.002)
    osc = OracleScene(sd)
    cam = synth.cerealbar_camera(48, 27)
    rp = render_params(spp_min=4, shading_mode=_ffi.SHADING_LEGACY)
    samples = osc.render_samples(cam, rp)
    img = osc.render(cam, rp)
    at = _ffi.ShadePoint()
    at.P[:] = (0.9, 0.35, 0.2); at.N[:] = (0.0, 1.0, 0.0)
    at.obj, at.bounce, at.diffuse_depth, at.pixel, at.sample, at.node = -1, 1, 0, 77, 1, 1
    g = np.random.default_rng(7)
    d = g.normal(size=(512, 3)).astype(np.float32); d[:, 1] = np.abs(d[:, 1]) + 0.05
    sec = osc.trace_secondary(at, d, recursive=True, child=1)
    np.savez_compressed(os.path.join(out, "legacy_small.npz"), samples=samples, linear=img["linear"],
                        rays=np.array([img["rays"][k] for k in ("primary", "bounce", "shadow")], np.int64), dirs=d, secondary=sec)


if __name__ == "__main__":
    if sys.argv[1:] == ["legacy"]:
        legacy_fixture()
    else:
        all_fixtures()
        legacy_fixture()
