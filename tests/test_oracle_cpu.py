"""CPU tests of the oracle (the CPU restatement of the reference hot path).

The reference holds no tests or golden vectors and cannot be built in this image, so the oracle is
pinned two ways: (1) analytic known answers computed here independently in float64 / pure Python, and (2) the
committed fixtures under tests/golden (oracle outputs on seeded inputs, tools/make_golden.py) as regression vectors.
"""
import ctypes as C
import os

import numpy as np
import pytest

from raytracer_amd import _ffi, synth
from raytracer_amd.scene import SceneData, Mesh, Material, Light, render_params, make_camera
from tests.oracle_ffi import OracleScene, OraclePhotons, lib
from tests.helpers import random_rays

GOLD = os.path.join(os.path.dirname(__file__), "golden")
ray_dt, hit_dt, _ = _ffi.np_dtypes()


def gold(name):
    return np.load(os.path.join(GOLD, name))


# ---- analytic known answers ------------------------------------------------------------------------------------
def test_box_slab_against_float64():
    """scene_box.cpp:8-46 against an independent float64 slab test (away from the rounding boundary)."""
    g = gold("box_kat.npz")
    boxes, rays = g["boxes"].astype(np.float64), g["rays"]
    o, d, tmax = rays["o"].astype(np.float64), rays["d"].astype(np.float64), rays["tmax"].astype(np.float64)
    exp = np.zeros(len(o), bool)
    margin = np.zeros(len(o))
    for i in range(len(o)):
        tmin, tm, ok = 0.0, tmax[i], True
        for a in range(3):
            if abs(d[i, a]) < 1e-20:
                if o[i, a] < boxes[i, a] or o[i, a] > boxes[i, 3 + a]:
                    ok = False
                    break
                continue
            t0, t1 = (boxes[i, a] - o[i, a]) / d[i, a], (boxes[i, 3 + a] - o[i, a]) / d[i, a]
            if t0 > t1:
                t0, t1 = t1, t0
            tmin, tm = max(tmin, t0), min(tm, t1)
        exp[i] = ok and tm >= tmin and tm >= 0
        margin[i] = abs(tm - tmin)
    L = lib()
    res = np.zeros(len(o), np.uint8)
    L.orc_box_intersect(g["boxes"].ctypes.data_as(C.c_void_p), rays.ctypes.data_as(C.c_void_p), C.c_size_t(len(o)),
                        res.ctypes.data_as(C.c_void_p))
    clear = margin > 1e-4
    assert clear.mean() > 0.9
    assert (res.astype(bool)[clear] == exp[clear]).all()
    assert 0.05 < exp.mean() < 0.95
    assert (res == g["hit"]).all()  # regression vector


def test_hits_well_inside_a_box_pass_the_reference_box_test():
    """The box-cull check of RTX_AMD_BOX_CULLS (raytracer_amd/csrc/traverse.h: well_inside_box, ref_reaches_triangle)
    accepts a hit without running the reference's slab test when the hit point lies inside the box by a margin of
    1e-3 of the extent + 1e-5 (|p| + t) per axis. That shortcut must never accept what scene_box.cpp:8-46 rejects: two
    million random boxes (flat to room-sized), rays from every direction and distance, segment ends just behind the hit."""
    rng = np.random.default_rng(2024)
    n = 2_000_000
    f = np.float32
    ext = (10.0 ** rng.uniform(-4, 1, (n, 3))).astype(f)
    ctr = rng.uniform(-5, 5, (n, 3)).astype(f)
    lo, hi = (ctr - ext * f(0.5)).astype(f), (ctr + ext * f(0.5)).astype(f)
    p = (lo + (hi - lo) * rng.uniform(0, 1, (n, 3)).astype(f)).astype(f)
    d = rng.normal(size=(n, 3))
    d[rng.random(n) < 0.1, rng.integers(0, 3)] = 0.0                       # axis-parallel rays too
    d = (d / np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-30)).astype(f)
    t = (10.0 ** rng.uniform(-3, 2.5, n)).astype(f)
    o = (p - d * t[:, None]).astype(f)
    tmax = np.where(rng.random(n) < 0.5, f(1e30), t * rng.uniform(1.0011, 2.0, n).astype(f)).astype(f)
    q = (o + d * t[:, None]).astype(f)                                     # the hit point as the kernel forms it
    eps = (f(1e-3) * (hi - lo) + f(1e-5) * (np.abs(q) + np.abs(t)[:, None])).astype(f)
    sure = ((q > lo + eps) & (q < hi - eps)).all(1) & (t < f(1e6)) & (tmax > t * f(1.001))
    assert sure.sum() > 100_000
    rays = np.zeros(n, ray_dt)
    rays["o"], rays["d"], rays["tmax"] = o, d, tmax
    boxes = np.ascontiguousarray(np.concatenate([lo, hi], 1), f)
    res = np.zeros(n, np.uint8)
    lib().orc_box_intersect(boxes.ctypes.data_as(C.c_void_p), rays.ctypes.data_as(C.c_void_p), C.c_size_t(n), res.ctypes.data_as(C.c_void_p))
    assert res[sure].all(), int((~res[sure].astype(bool)).sum())
    assert not res.all()                                                   # the unsure rest does hold rejected hits (thin boxes, rims)


def test_triangle_hits_against_float64():
    """objects.cpp:140-171: t/u/v of the oracle against a float64 Moller-Trumbore on the same triangle."""
    sd, cam = synth.single_mesh_scene("teapots", 0.02)
    osc = OracleScene(sd)
    rays = random_rays(np.random.default_rng(1), 5000, osc.bounds())
    h = osc.trace_closest(rays, _ffi.HIT_FRONT_AND_BACK)
    m = sd.meshes[0]
    hit = h["obj"] >= 0
    assert hit.sum() > 300
    f = m.f[h["face"][hit]]
    v0, v1, v2 = (m.v[f[:, k]].astype(np.float64) for k in range(3))
    o, d = rays["o"][hit].astype(np.float64), rays["d"][hit].astype(np.float64)
    e1, e2 = v1 - v0, v2 - v0
    p = np.cross(d, e2)
    det = (e1 * p).sum(1)
    tv = o - v0
    u = (tv * p).sum(1) / det
    q = np.cross(tv, e1)
    v = (d * q).sum(1) / det
    t = (e2 * q).sum(1) / det
    ok = np.abs(det) > 1e-4
    assert np.abs(h["t"][hit][ok] - t[ok]).max() <= 2e-5 * np.abs(t[ok]).max()
    assert np.abs(h["bu"][hit][ok] - u[ok]).max() < 1e-3 and np.abs(h["bv"][hit][ok] - v[ok]).max() < 1e-3
    # hit point lies on the ray, front flag is the sign of dir . GN
    assert np.abs(h["p"][hit] - (o + d * h["t"][hit][:, None])).max() < 1e-5
    gn = np.cross(e1, e2)
    assert ((np.sign((d * gn).sum(1)) < 0) == (h["front"][hit] == 1))[ok].all()


def test_bvh_equals_bruteforce_and_hitside():
    """The reference's own A/B check (--nobvh, objects.cpp:256-265) on the oracle, for all three hitSide modes."""
    sd, cam = synth.single_mesh_scene("teapots", 0.01)
    osc = OracleScene(sd)
    rays = random_rays(np.random.default_rng(2), 3000, osc.bounds())
    for side in (_ffi.HIT_FRONT, _ffi.HIT_BACK, _ffi.HIT_FRONT_AND_BACK):
        a = osc.trace_closest(rays, side, use_bvh=True)
        b = osc.trace_closest(rays, side, use_bvh=False)
        assert (a["face"] == b["face"]).all() and (a["t"] == b["t"]).all()
    both = osc.trace_closest(rays, _ffi.HIT_FRONT_AND_BACK)
    fr = osc.trace_closest(rays, _ffi.HIT_FRONT)
    bk = osc.trace_closest(rays, _ffi.HIT_BACK)
    hit = both["obj"] >= 0
    # the closest two-sided hit is the closer of the one-sided hits
    assert np.array_equal(both["t"][hit], np.minimum(fr["t"], bk["t"])[hit])
    assert (fr["front"][fr["obj"] >= 0] == 1).all() and (bk["front"][bk["obj"] >= 0] == 0).all()


def _py_splitmix(x):
    M = (1 << 64) - 1
    x = (x + 0x9E3779B97F4A7C15) & M
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
    return x, z ^ (z >> 31)


def test_sampler_known_answers():
    """Halton (radical inverse), PixelShifts (splitmix64, RayCaster.cpp:65-81) and the counter RNG range."""
    L = lib()
    for i in (1, 2, 3, 5, 6, 11, 1000, 123457):
        assert L.orc_halton(i, 2) == pytest.approx(int(format(i, "b")[::-1], 2) / (1 << i.bit_length()), abs=1e-6)
        for b in (3, 5, 7):
            r, f, k = 0.0, 1.0 / b, i
            while k:
                r += f * (k % b)
                k //= b
                f /= b
            assert L.orc_halton(i, b) == pytest.approx(r, abs=2e-6)
    assert L.orc_halton(0, 2) == 0.0
    # splitmix64 reference sequence (Vigna) from state 1234567: first output 6457827717110365317
    s, z = _py_splitmix(1234567)
    assert z == 6457827717110365317
    for x, y in ((0, 0), (17, 5), (1919, 1079)):
        s = (x << 32) | y
        s, h1 = _py_splitmix(s)
        s, h2 = _py_splitmix(s)
        a, b = C.c_float(), C.c_float()
        L.orc_pixel_shifts(x, y, C.byref(a), C.byref(b))
        assert a.value == np.float32((h1 & 0xFFFFFFFF) / 4294967296.0) and b.value == np.float32((h2 & 0xFFFFFFFF) / 4294967296.0)
    u = np.array([L.orc_rnd(7, a, b, s, d) for a in range(8) for b in range(8) for s in range(4) for d in range(8)])
    assert u.min() >= 0.0 and u.max() < 1.0 and abs(u.mean() - 0.5) < 0.03 and len(np.unique(u)) > 0.99 * len(u)
    g = gold("sampler_kat.npz")
    assert np.array_equal(np.array([[L.orc_halton(int(i), b) for b in (2, 3, 5, 7)] for i in g["idx"]], np.float32), g["halton"])


def test_filters_and_color24():
    """RayCaster.cpp:643-690 and :33-48."""
    L = lib()
    assert L.orc_filter1d(_ffi.FILTER_TENT, 1.0, 0.0) == 1.0 and L.orc_filter1d(_ffi.FILTER_TENT, 1.0, 1.0) == 0.0
    assert L.orc_filter1d(_ffi.FILTER_TENT, 1.0, -0.25) == 0.75
    assert L.orc_filter1d(_ffi.FILTER_TENT, 0.0, 0.5) == 0.5          # radius <= 0 -> default support 1
    assert L.orc_filter1d(_ffi.FILTER_MITCHELL, 0.0, 0.0) == pytest.approx((6 - 2 / 3) / 6, rel=1e-6)
    assert L.orc_filter1d(_ffi.FILTER_MITCHELL, 0.0, 2.0) == 0.0
    assert L.orc_filter1d(_ffi.FILTER_LANCZOS, 2.0, 0.0) == 1.0 and abs(L.orc_filter1d(_ffi.FILTER_LANCZOS, 2.0, 1.0)) < 1e-6
    assert L.orc_filter1d(_ffi.FILTER_GAUSSIAN, 1.5, 0.75) == pytest.approx(np.exp(-0.5), rel=1e-6)
    out = (C.c_ubyte * 3)()
    L.orc_to_color24((C.c_float * 3)(0.5, 2.0, -1.0), 0, out)
    assert tuple(out) == (127, 255, 0)                                # truncation, clamp
    L.orc_to_color24((C.c_float * 3)(0.5, 0.0031308, 0.001), 1, out)
    assert out[0] == int((1.055 * 0.5 ** 0.41666 - 0.055) * 255) and out[2] == int(0.001 * 12.92 * 255)


def _one_triangle_scene(material):
    sd = SceneData()
    sd.materials = [material]
    v = np.array([[-5, 0, -5], [5, 0, -5], [5, 0, 5], [-5, 0, 5]], np.float32)
    sd.meshes = [Mesh(v, [[0, 2, 1], [0, 3, 2]], vn=np.tile([[0, 1, 0]], (4, 1)), material=0)]
    return sd


def test_direct_light_analytic():
    """Point light (size 0) over a diffuse plane, one bounce: L = kd/pi * I/d^2 * cos (Shading.cpp:125-133, :305-325,
    RayCaster.cpp:498-502 with MIS weight 1/(1+pdf_bsdf^2))."""
    kd = 0.6
    sd = _one_triangle_scene(Material(diffuse=(kd, kd, kd), specular=(0, 0, 0), glossiness=10.0))
    sd.lights = [Light((0, 2, 0), (4, 4, 4), size=0.0, attenuation=1.0)]
    cam = make_camera((0, 3, 0.001), (0, 0, 0), (0, 0, -1), fov=20, width=8, height=8)
    osc = OracleScene(sd)
    s = osc.render_samples(cam, render_params(spp_min=1, max_bounces=1))
    # pixel near the centre looks at ~(0,0,0): cos = 1, d = 2
    pdf_b = 1.0 / np.pi
    expect = (4.0 / 4.0) * (kd / np.pi) * (1.0 / (1.0 + pdf_b ** 2))
    assert s[4, 4, 0, 0] == pytest.approx(expect, rel=2e-2)
    assert s[4, 4, 0, 3] == pytest.approx(3.0, rel=1e-2)              # primaryDist


def test_glass_slab_visibility_chain():
    """Visible() (RayCaster.cpp:193-277): two glass interfaces attenuate by (1 - F0)^2 at normal incidence; an opaque
    sheet blocks; the boolean variant treats glass as a blocker."""
    glass = Material(diffuse=(0, 0, 0), specular=(0, 0, 0), refraction=(1, 1, 1), ior=1.5, glossiness=1e5)
    sd = _one_triangle_scene(glass)
    v = sd.meshes[0].v
    sd.meshes.append(Mesh(v + np.array([0, 0.5, 0], np.float32), [[0, 1, 2], [0, 2, 3]], material=0))
    osc = OracleScene(sd)
    rays = np.zeros(1, ray_dt)
    rays["o"], rays["d"], rays["tmax"] = (0.1, -1.0, 0.2), (0, 1, 0), 3.0
    n = np.array([[0, 1, 0]], np.float32)
    F0 = ((1 - 1.5) / (1 + 1.5)) ** 2
    assert osc.trace_visibility(rays, n, True)[0] == pytest.approx((1 - F0) ** 2, rel=1e-5)
    assert osc.trace_visibility(rays, n, False)[0] == 0.0
    rays["tmax"] = 0.5
    assert osc.trace_visibility(rays, n, True)[0] == 1.0             # segment ends before the first sheet
    sd.materials.append(Material(diffuse=(0.5, 0.5, 0.5)))
    sd.meshes[1].material = 1
    rays["tmax"] = 3.0
    assert OracleScene(sd).trace_visibility(rays, n, True)[0] == 0.0  # opaque second sheet


def test_bsdf_sampling_consistency():
    """GenerateSample / GetSampleInfo (Shading.cpp:280-713): for the non-delta lobes the evaluated pdf of the sampled
    direction is at least the sampled lobe's pdf, weights are finite and non-negative."""
    g = gold("bsdf_kat.npz")
    sd = synth.build_scene(0.002)
    osc = OracleScene(sd)
    L = lib()
    ctx, so = g["ctx"], g["sample"]
    out = np.zeros(6, np.float32)
    checked = 0
    for i in range(len(ctx)):
        if so[i, 9] == 0 or so[i, 7] < 0 or so[i, 8] == 4:  # failed, delta (TIR) or transmission lobe
            continue
        m = i // 64
        L.orc_bsdf_eval(osc.h, m, ctx[i].ctypes.data_as(C.c_void_p), so[i, 0:3].copy().ctypes.data_as(C.c_void_p),
                        out.ctypes.data_as(C.c_void_p))
        assert np.isfinite(so[i]).all() and (so[i, 3:6] >= 0).all() and so[i, 6] > 0
        assert out[3] >= so[i, 6] * (1 - 0.05)  # pow(NoH, gloss) amplifies fp32 rounding of the half vector for gloss >= 1000
        checked += 1
    assert checked > 300


def test_photon_estimate_against_numpy():
    """The k-nearest cone-filtered estimate (stand-in definition) against a brute-force numpy evaluation."""
    sd = synth.build_scene(0.002)
    osc = OracleScene(sd)
    ph = OraclePhotons(osc, 60000)
    info = ph.info()
    assert info["n_global"] > 20 and info["n_caustic"] > 20
    for which, radius in ((0, info["photon_radius"]), (1, info["caustic_radius"])):
        p = ph.download(which)                                        # scaled powers after finalize
        probe = p["pos"][:40] + np.float32(0.01)
        nrm = np.tile(np.array([[0, 1, 0]], np.float32), (len(probe), 1))
        est = ph.estimate(which, probe, nrm)
        for j in range(len(probe)):
            d2 = ((p["pos"].astype(np.float64) - probe[j]) ** 2).sum(1)
            idx = np.argsort(d2, kind="stable")
            idx = idx[d2[idx] <= radius * radius][:100]
            rr2 = d2[idx[-1]] if len(idx) == 100 else radius * radius
            w = 1 - np.sqrt(d2[idx]) / np.sqrt(rr2)
            keep = (p["dir"][idx] @ nrm[j]) < 0
            ref = (p["power"][idx][keep] * w[keep, None]).sum(0) * 3 / (np.pi * rr2)
            assert np.allclose(est[j], ref, rtol=2e-3, atol=1e-9)


# ---- regression vectors (tests/golden) ---------------------------------------------------------------------------
def test_golden_closest_hit():
    g = gold("closest_c1.npz")
    sd, cam = synth.single_mesh_scene("teapots", 0.02)
    cam.width = cam.height = 64
    osc = OracleScene(sd)
    prim, t, front = osc.render_primary(cam, _ffi.HIT_FRONT_AND_BACK)
    assert np.array_equal(prim, g["prim"]) and np.array_equal(t, g["t"]) and np.array_equal(front, g["front"])
    h = osc.trace_closest(g["rays"], _ffi.HIT_FRONT_AND_BACK)
    for k in ("t", "obj", "face", "p", "N", "GN", "uvw", "dpdu", "dpdv", "front"):
        assert np.array_equal(h[k], g["hits"][k]), k


def test_golden_path_and_adaptive():
    g = gold("path_small.npz")
    sd = synth.build_scene(0.002)
    osc = OracleScene(sd)
    cam = synth.cerealbar_camera(48, 27)
    s = osc.render_samples(cam, render_params(spp_min=4, max_bounces=64))
    assert np.allclose(s, g["samples"], rtol=1e-6, atol=1e-7)
    img = osc.render(cam, render_params(spp_min=4, max_bounces=64))
    assert np.allclose(img["linear"], g["linear"], rtol=1e-5, atol=1e-6) and np.array_equal(img["rgb8"], g["rgb8"])
    assert [img["rays"][k] for k in ("primary", "bounce", "shadow")] == list(g["rays"])
    ada = osc.render(cam, render_params(spp_min=8, spp_max=32, max_bounces=8, ci_halfwidth=0.02))
    assert np.array_equal(ada["samples"], g["adaptive_samples"])
    assert set(np.unique(ada["samples"])) <= {8, 16, 24, 32} and (ada["samples"] > 8).any()
    # threads: per-thread films merged (mergeToImage) give the single-thread image up to float association
    img4 = osc.render(cam, render_params(spp_min=4, max_bounces=64), nthreads=4)
    assert np.allclose(img4["linear"], img["linear"], rtol=1e-4, atol=1e-5) and np.array_equal(img4["samples"], img["samples"])


def test_golden_photons():
    g = gold("photon_small.npz")
    sd = synth.build_scene(0.002)
    osc = OracleScene(sd)
    ph = OraclePhotons(osc, 100000)
    info = ph.info()
    assert info["n_global"] == int(g["n_global"]) and info["n_caustic"] == int(g["n_caustic"])
    assert np.allclose([info["photon_radius"], info["caustic_radius"]], g["info_radii"])
    nn = np.tile(np.array([[0, 1, 0]], np.float32), (len(g["probe"]), 1))
    assert np.allclose(ph.estimate(0, g["probe"], nn), g["est_global"], rtol=1e-5, atol=1e-12)
    assert np.allclose(ph.estimate(1, g["probe"], nn), g["est_caustic"], rtol=1e-5, atol=1e-12)
    # photon-id striding (the multi-GPU split) partitions the same photon set
    a = OraclePhotons(osc, 100000, first_photon=0, photon_stride=2, finalize=False)
    b = OraclePhotons(osc, 100000, first_photon=1, photon_stride=2, finalize=False)
    assert a.info()["n_global"] + b.info()["n_global"] == info["n_global"]
    assert a.info()["n_caustic"] + b.info()["n_caustic"] == info["n_caustic"]


def test_edge_cases():
    """Empty scene, empty ray batch, degenerate triangles, zero-length direction: 'no hit' rather than failure
    (SURVEY.md §8b error conventions)."""
    empty = OracleScene(SceneData())
    rays = np.zeros(4, ray_dt)
    rays["d"], rays["tmax"] = (0, 0, 1), 1e30
    assert (empty.trace_closest(rays)["obj"] == -1).all()
    assert empty.trace_closest(rays[:0]).shape[0] == 0
    sd = SceneData()
    sd.materials = [Material()]
    sd.meshes = [Mesh(np.zeros((3, 3), np.float32), [[0, 1, 2]], material=0)]      # zero-area triangle
    deg = OracleScene(sd)
    assert (deg.trace_closest(rays, _ffi.HIT_FRONT_AND_BACK)["obj"] == -1).all()
    rays["d"] = 0
    assert (deg.trace_closest(rays, _ffi.HIT_FRONT_AND_BACK)["obj"] == -1).all()


def test_exact_ties_take_the_smaller_face_index():
    """Faces at exactly the same t: the brute-force loop of objects.cpp:257-264 keeps the first (t >= hInfo.z rejects);
    the BVH walk must give the same answer whatever order it visits them in."""
    rng = np.random.default_rng(11)
    n = 40
    ctr = rng.uniform(-1, 1, (n, 1, 3))
    v = (ctr + rng.normal(scale=0.4, size=(n, 3, 3))).reshape(-1, 3).astype(np.float32)
    f = np.arange(3 * n, dtype=np.int32).reshape(n, 3)
    f = np.concatenate([f, f[:10], f[5:15]])                    # exact duplicates of faces 0..14, some of them twice
    sd = SceneData()
    sd.materials = [Material()]
    sd.meshes = [Mesh(v, f, material=0)]
    osc = OracleScene(sd)
    ray_dt, _, _ = _ffi.np_dtypes()
    rays = np.zeros(4000, dtype=ray_dt)
    rays["o"] = rng.uniform(-3, 3, (4000, 3)).astype(np.float32)
    d = rng.uniform(-1, 1, (4000, 3)).astype(np.float32) - rays["o"]
    rays["d"] = d / np.linalg.norm(d, axis=1, keepdims=True)
    rays["tmax"] = 1e30
    for side in (_ffi.HIT_FRONT, _ffi.HIT_BACK, _ffi.HIT_FRONT_AND_BACK):
        a = osc.trace_closest(rays, side, use_bvh=True)
        b = osc.trace_closest(rays, side, use_bvh=False)
        assert (a["face"] == b["face"]).all() and (a["t"] == b["t"]).all()
        assert (a["face"] < n).all() and (a["face"][a["obj"] >= 0] < 15).sum() > 100


def test_mirrored_transform_facing_and_tangent_frame():
    """Independent pins for the oracle's transform handling (IntersectNodeAdapter.h:62,93): a closed box under a
    mirroring transform, seen from outside, shows front faces only, with outward geometric normals; a face without
    usable texture coordinates carries the object-space fallback frame (objects.cpp:233-240) to the world as
    M * tangent, M * bitangent."""
    M = np.array([[-0.7, 0.1, 0.0], [0.1, 0.8, 0.2], [0.0, -0.3, 0.6]], np.float64)
    T = np.array([0.2, 0.9, -0.4], np.float64)
    assert np.linalg.det(M) < 0
    xf = np.concatenate([M, T[:, None]], 1).astype(np.float32).reshape(12)
    c = np.array([[i & 1, (i >> 1) & 1, (i >> 2) & 1] for i in range(8)], np.float32) * 2 - 1
    f = [[0, 2, 1], [1, 2, 3], [4, 5, 6], [5, 7, 6], [0, 1, 4], [1, 5, 4], [2, 6, 3], [3, 6, 7], [0, 4, 2], [2, 4, 6], [1, 3, 5], [3, 7, 5]]
    sd = SceneData()
    sd.materials = [Material()]
    sd.meshes = [Mesh(c, f, xform=xf, material=0)]
    osc = OracleScene(sd)
    rng = np.random.default_rng(9)
    ray_dt, _, _ = _ffi.np_dtypes()
    n = 2000
    rays = np.zeros(n, dtype=ray_dt)
    o = rng.normal(size=(n, 3))
    o = o / np.linalg.norm(o, axis=1, keepdims=True) * 6.0 + T
    tgt = (M @ rng.uniform(-0.8, 0.8, (n, 3)).T).T + T                 # points inside the box
    d = tgt - o
    rays["o"], rays["d"], rays["tmax"] = o.astype(np.float32), (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32), 1e30
    h = osc.trace_closest(rays, _ffi.HIT_FRONT_AND_BACK)
    assert (h["obj"] == 0).all() and (h["front"] == 1).all()
    centre_to_hit = h["p"].astype(np.float64) - T
    assert (np.einsum("ij,ij->i", centre_to_hit, h["GN"].astype(np.float64)) > 0).all()      # outward
    assert (np.einsum("ij,ij->i", rays["d"].astype(np.float64), h["GN"].astype(np.float64)) < 0).all()
    assert (osc.trace_closest(rays, _ffi.HIT_BACK)["obj"] >= 0).all()                          # the far side of the box
    assert np.array_equal(osc.trace_closest(rays, _ffi.HIT_FRONT)["face"], h["face"])
    # fallback tangent frame: no texture coordinates -> tangent from the object-space normal, then through M
    Minv = np.linalg.inv(M)
    for i in range(0, n, 97):
        n_obj = M.T @ h["GN"][i].astype(np.float64)
        n_obj /= np.linalg.norm(n_obj)
        t = np.array([-n_obj[1], n_obj[0], 0.0]) if abs(n_obj[0]) > 0.5 else np.array([0.0, -n_obj[2], n_obj[1]])
        t /= np.linalg.norm(t)
        b = np.cross(n_obj, t)
        b /= np.linalg.norm(b)
        assert np.allclose(h["dpdu"][i], M @ t, atol=2e-5) and np.allclose(h["dpdv"][i], M @ b, atol=2e-5)
        assert np.allclose(Minv @ h["dpdu"][i].astype(np.float64), t, atol=5e-5)


def test_fuzz_generator_is_deterministic():
    """The regression seeds of tests/test_fuzz_gpu.py name scenes: the generator must give the same scene for the same seed."""
    from tools.fuzz_parity import rand_scene, spice_scene
    for seed in (5, 39, 287):
        a, cam_a, _ = rand_scene(seed)
        b, cam_b, _ = rand_scene(seed)
        spice_scene(a, cam_a, seed)
        spice_scene(b, cam_b, seed)
        assert len(a.meshes) == len(b.meshes) and (cam_a.width, cam_a.height) == (cam_b.width, cam_b.height)
        for ma, mb in zip(a.meshes, b.meshes):
            assert np.array_equal(ma.v, mb.v) and np.array_equal(ma.f, mb.f) and np.array_equal(ma.xform, mb.xform) and ma.material == mb.material
        assert [vars(m) for m in a.materials] == [vars(m) for m in b.materials]
    sd, cam, _ = rand_scene(39)                                        # the seed that found the mirrored-transform defect
    assert any(np.linalg.det(m.xform.reshape(3, 4)[:, :3]) < 0 for m in sd.meshes if m.kind == _ffi.OBJ_MESH)


def test_photon_map_capacity():
    """PhotonMapBuilder.cpp:27-28: the caustics map holds numPhotons / 10 records, the global map numPhotons; a full map
    takes no more (stand-in for the absent cyPhotonMap): the records emitted first stay, and the caustic scale is
    1 / (records kept)."""
    from tests.oracle_ffi import OraclePhotons, OracleScene
    from tests.test_photon_gpu import _glass_over_floor
    osc = OracleScene(_glass_over_floor())
    N = 20000
    raw = OraclePhotons(osc, N, finalize=False)
    full = raw.download(1)
    assert len(full) > N // 10
    kept = OraclePhotons(osc, N).download(1)
    assert len(kept) == N // 10
    assert np.array_equal(kept["pos"], full["pos"][:N // 10])
    assert np.array_equal(kept["power"], full["power"][:N // 10] * (np.float32(1.0) / np.float32(N // 10)))
