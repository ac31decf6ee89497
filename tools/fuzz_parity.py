#!/usr/bin/env python3
"""Differential fuzzing of the HIP path against the oracle: random small scenes (triangle soups, boxes, analytic spheres and
planes under random affine transforms; Blinn / Phong / microfacet materials with random reflection, refraction,
absorption, emission, textures and bump maps; 0..8 point and sphere lights; thin-lens cameras), rendered through the C
ABI and through oracle/liboracle.so with the same parameters, per-sample radiance compared.

    python tools/fuzz_parity.py [--seeds 0:40] [--mode path|legacy|both] [--photons]

Prints one line per scene and exits non-zero if a scene falls below --min-agree. TEST INFRASTRUCTURE (uses the oracle).
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rtx  # noqa: E402
from raytracer_amd import _ffi  # noqa: E402
from raytracer_amd.scene import Light, Material, Mesh, SceneData, TexMap, make_camera, render_params  # noqa: E402
from tests.oracle_ffi import OraclePhotons, OracleScene  # noqa: E402


def rand_xform(rng, scale=1.0, spread=2.0, uniform=False):
    A = rng.normal(size=(3, 3))
    Q, _ = np.linalg.qr(A)
    s = rng.uniform(0.3, 1.2, 3) * scale
    if uniform:
        s[:] = s[0]
    M = Q @ np.diag(s)
    if rng.random() < 0.3:
        M[:, 0] = -M[:, 0]                                   # mirrored instance: flips the winding
    t = rng.uniform(-spread, spread, 3)
    t[1] = abs(t[1]) * 0.6
    return np.concatenate([M, t[:, None]], axis=1).astype(np.float32).reshape(12)


def box_mesh(rng, material, xform):
    c = np.array([[i & 1, (i >> 1) & 1, (i >> 2) & 1] for i in range(8)], np.float32) * 2 - 1
    f = [[0, 2, 1], [1, 2, 3], [4, 5, 6], [5, 7, 6], [0, 1, 4], [1, 5, 4], [2, 6, 3], [3, 6, 7], [0, 4, 2], [2, 4, 6], [1, 3, 5], [3, 7, 5]]
    vt = np.concatenate([c[:, :2] * 0.5 + 0.5, np.zeros((8, 1), np.float32)], axis=1).astype(np.float32)
    return Mesh(c, f, vt=vt, xform=xform, material=material)


def soup_mesh(rng, material, n, xform):
    ctr = rng.uniform(-1, 1, (n, 1, 3))
    v = (ctr + rng.normal(scale=0.35, size=(n, 3, 3))).reshape(-1, 3).astype(np.float32)
    f = np.arange(3 * n, dtype=np.int32).reshape(n, 3)
    vn = None
    if rng.random() < 0.5:                                   # smooth (and deliberately not unit-length) normals
        vn = rng.normal(size=(3 * n, 3)).astype(np.float32)
    vt = rng.uniform(0, 1, (3 * n, 3)).astype(np.float32) if rng.random() < 0.7 else None
    return Mesh(v, f, vn=vn, vt=vt, xform=xform, material=material)


def rand_material(rng, n_tex):
    kind = rng.choice(["diffuse", "glossy", "mirror", "glass", "emissive", "micro"], p=[0.3, 0.2, 0.12, 0.2, 0.06, 0.12])
    tex = lambda p=0.35: int(rng.integers(0, n_tex)) if (n_tex and rng.random() < p) else -1
    c = lambda lo, hi: tuple(rng.uniform(lo, hi, 3))
    mtype = int(rng.choice([_ffi.MTL_BLINN, _ffi.MTL_PHONG]))
    if kind == "diffuse":
        return Material(type=mtype, diffuse=c(0.2, 0.9), specular=c(0, 0.3), glossiness=float(rng.uniform(5, 80)), diffuse_tex=tex(0.5),
                        bump_tex=tex(0.3), bump_scale=float(rng.uniform(0.2, 2.0)))
    if kind == "glossy":
        return Material(type=mtype, diffuse=c(0.05, 0.5), specular=c(0.3, 0.9), glossiness=float(rng.uniform(20, 2000)), specular_tex=tex(),
                        diffuse_tex=tex())
    if kind == "mirror":
        return Material(type=mtype, diffuse=c(0, 0.2), specular=c(0.2, 0.8), reflection=c(0.5, 0.95), glossiness=float(rng.choice([50.0, 1e3, 1e5])),
                        reflection_tex=tex(0.2))
    if kind == "glass":
        return Material(type=mtype, diffuse=c(0, 0.1), specular=c(0.3, 0.9), refraction=c(0.6, 0.98), ior=float(rng.uniform(1.05, 2.2)),
                        glossiness=float(rng.choice([100.0, 1e4])), absorption=c(0, 1.5), reflection=c(0, 0.3) if rng.random() < 0.3 else (0, 0, 0),
                        refraction_tex=tex(0.15))
    if kind == "emissive":
        return Material(type=mtype, diffuse=c(0.1, 0.6), emission=c(0.2, 3.0), emission_tex=tex(0.4))
    return Material(type=_ffi.MTL_MICROFACET, diffuse=c(0.1, 0.9), specular=(float(rng.uniform(0.02, 0.9)), 0, 0),
                    reflection=(float(rng.choice([0.0, 1.0, rng.uniform(0, 1)])), 0, 0),
                    refraction=c(0, 0.9) if rng.random() < 0.3 else (0, 0, 0), ior=float(rng.uniform(1.2, 1.9)), absorption=c(0, 1.0),
                    diffuse_tex=tex(), specular_tex=tex(0.2), reflection_tex=tex(0.2), bump_tex=tex(0.2))


def rand_scene(seed):
    rng = np.random.default_rng(seed)
    sd = SceneData()
    n_tex = int(rng.integers(0, 4))
    for _ in range(n_tex):
        h, w = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        if rng.random() < 0.5:
            t = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
        else:
            t = rng.uniform(0, 2.0, (h, w, 3)).astype(np.float32)
        sd.texmaps.append(TexMap(t, xform=rand_xform(rng, 2.0, 0.5) if rng.random() < 0.4 else None))
    sd.materials = [rand_material(rng, n_tex) for _ in range(int(rng.integers(1, 7)))]
    nm = len(sd.materials)
    mat = lambda: int(rng.integers(-1 if rng.random() < 0.05 else 0, nm))
    floor = np.array([[-5, 0, -5], [5, 0, -5], [5, 0, 5], [-5, 0, 5]], np.float32)
    if rng.random() < 0.8:
        sd.meshes.append(Mesh(floor, [[0, 2, 1], [0, 3, 2]], vt=(floor[:, [0, 2, 1]] * 0.1 + 0.5).astype(np.float32), material=mat()))
    for _ in range(int(rng.integers(1, 9))):
        k = rng.random()
        if k < 0.3:
            sd.meshes.append(box_mesh(rng, mat(), rand_xform(rng, 0.6)))
        elif k < 0.55:
            sd.meshes.append(soup_mesh(rng, mat(), int(rng.integers(1, 60)), rand_xform(rng, 0.8) if rng.random() < 0.7 else None))
        elif k < 0.85:
            sd.meshes.append(Mesh.sphere(xform=rand_xform(rng, 0.7, uniform=rng.random() < 0.5), material=mat()))
        else:
            sd.meshes.append(Mesh.plane(xform=rand_xform(rng, 2.0), material=mat()))
    for _ in range(int(rng.choice([0, 1, 2, 3, 8], p=[0.08, 0.3, 0.3, 0.27, 0.05]))):
        pos = rng.uniform(-4, 4, 3)
        pos[1] = rng.uniform(1.5, 5)
        sd.lights.append(Light(tuple(pos), tuple(rng.uniform(2, 25, 3)), size=float(rng.choice([0.0, rng.uniform(0.05, 0.8)])),
                               attenuation=float(rng.choice([1.0, 1.0, rng.uniform(0.2, 2.0)]))))
    sd.environment = tuple(rng.uniform(0, 0.5, 3)) if rng.random() < 0.8 else (0, 0, 0)
    if n_tex and rng.random() < 0.4:
        sd.environment_tex = int(rng.integers(0, n_tex))
    if rng.random() < 0.3:
        sd.background = tuple(rng.uniform(0, 1, 3))
        sd.background_emission = float(rng.choice([0.0, 0.5]))
        if n_tex and rng.random() < 0.5:
            sd.background_tex = int(rng.integers(0, n_tex))
    eye = rng.uniform(-1, 1, 3) + np.array([0, 1.8, 5.0])
    cam = make_camera(tuple(eye), tuple(rng.uniform(-0.5, 0.5, 3) + np.array([0, 0.5, 0])), fov=float(rng.uniform(25, 70)),
                      focaldist=float(rng.uniform(3, 7)), dof=float(rng.choice([0.0, rng.uniform(0.01, 0.2)])),
                      width=int(rng.integers(8, 48)), height=int(rng.integers(8, 32)))
    return sd, cam, rng


def spice_scene(sd, cam, seed):
    """Second-stage mutations from a stream of their own (the base scene of a seed stays what it was): separate normal /
    texture index arrays, per-face material ids, zero-area and duplicated triangles, overlapping coplanar faces,
    extreme material parameters, tiny textures, texture coordinates far outside [0, 1], a visible sphere light in front
    of the camera."""
    rng = np.random.default_rng(seed + 100003)
    mask = int(os.environ.get("FUZZ_SPICE_MASK", "255"))     # debugging aid: switch single mutation classes off
    for m in sd.meshes:
        if m.kind != _ffi.OBJ_MESH or m.nf == 0:
            continue
        nf = m.f.shape[0]
        if rng.random() < 0.5 and mask & 1:
            m.face_mtl = rng.integers(0, 5, nf).astype(np.int32)
        if m.vn is not None and rng.random() < 0.5 and mask & 2:
            m.fn = rng.integers(0, m.vn.shape[0], (nf, 3)).astype(np.int32)
        if m.vt is not None and rng.random() < 0.5 and mask & 4:
            m.ft = rng.integers(0, m.vt.shape[0], (nf, 3)).astype(np.int32)
            m.vt = (m.vt * rng.choice([1.0, 7.0, -3.0])).astype(np.float32)
        if rng.random() < 0.4 and mask & 8:                                   # zero-area faces and an exact duplicate of face 0
            extra = np.array([[m.f[0, 0], m.f[0, 0], m.f[0, 1]], m.f[0], [m.f[0, 0], m.f[0, 1], m.f[0, 1]]], np.int32)
            m.f = np.ascontiguousarray(np.concatenate([m.f, extra]))
            for name in ("fn", "ft"):
                a = getattr(m, name)
                if a is not None:
                    setattr(m, name, np.ascontiguousarray(np.concatenate([a, a[:1].repeat(3, 0)])))
            if m.face_mtl is not None:
                m.face_mtl = np.ascontiguousarray(np.concatenate([m.face_mtl, m.face_mtl[:1].repeat(3)]))
    boxes = [m for m in sd.meshes if m.kind == _ffi.OBJ_MESH and m.nf == 12]
    if boxes and rng.random() < 0.5 and mask & 16:                             # the same box again with another material: every face ties
        b = boxes[0]
        sd.meshes.append(Mesh(b.v.copy(), b.f.copy(), vt=None if b.vt is None else b.vt.copy(), xform=b.xform.copy(),
                              material=int(rng.integers(0, len(sd.materials)))))
    for mt in sd.materials:
        r = rng.random() if mask & 32 else 1.0
        if r < 0.15:
            mt.glossiness = float(rng.choice([0.0, 1.0, 1e6]))
        elif r < 0.3:
            mt.ior = float(rng.choice([1.0, 0.7, 3.5]))
        elif r < 0.4:
            mt.absorption = (0.0, 0.0, 0.0)
        elif r < 0.5 and mt.type != _ffi.MTL_MICROFACET:
            mt.reflection = tuple(rng.uniform(0.3, 0.6, 3))
            mt.refraction = tuple(rng.uniform(0.3, 0.6, 3))
    if sd.texmaps and rng.random() < 0.5 and mask & 64:
        sd.texmaps[0] = TexMap(rng.integers(0, 256, (1, 1, 3)).astype(np.uint8))
    if rng.random() < 0.5 and mask & 128:
        sd.lights = sd.lights[:7] + [Light((float(rng.uniform(-0.5, 0.5)), 1.2, 2.0), tuple(rng.uniform(5, 20, 3)), size=float(rng.uniform(0.2, 0.5)))]
    return sd, cam


def rand_rays(rng, n, segments=False):
    ray_dt, _, _ = _ffi.np_dtypes()
    r = np.zeros(n, dtype=ray_dt)
    o = rng.uniform(-4, 4, (n, 3)).astype(np.float32)
    o[:, 1] = np.abs(o[:, 1]) + 0.05
    tgt = rng.uniform(-2, 2, (n, 3)).astype(np.float32)
    tgt[:, 1] = np.abs(tgt[:, 1]) * 0.6
    d = tgt - o
    ln = np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    r["o"], r["d"] = o, d / ln
    r["tmax"] = (ln[:, 0] * rng.uniform(0.3, 1.5, n)).astype(np.float32) if segments else np.float32(1e30)
    axis = rng.random(n) < 0.05                                # a few axis-parallel directions
    r["d"][axis] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, int(axis.sum()))] * rng.choice([-1.0, 1.0], (int(axis.sum()), 1)).astype(np.float32)
    return r


def api_leg(gs, osc, sd, cam, rng, seed):
    """trace_closest / trace_visibility on random rays, and an adaptive full-frame render with a random filter."""
    msgs = []
    rays = rand_rays(rng, 3000)
    side = int(rng.choice([_ffi.HIT_FRONT, _ffi.HIT_BACK, _ffi.HIT_FRONT_AND_BACK]))
    wl = bool(rng.random() < 0.5)
    g, o = gs.trace_closest(rays, hit_side=side, with_lights=wl), osc.trace_closest(rays, hit_side=side, with_lights=wl)
    same = (g["obj"] == o["obj"]) & (g["face"] == o["face"]) & (g["light"] == o["light"]) & (g["mtl_id"] == o["mtl_id"])
    k = same & ((o["obj"] >= 0) | (o["light"] >= 0))
    ok_t = np.abs(g["t"][k] - o["t"][k]) <= 2e-5 * (1 + np.abs(o["t"][k]))
    ok_f = g["front"][k] == o["front"][k]
    ok_n = np.abs(g["N"][k] - o["N"][k]).max(-1) <= 1e-4 if k.any() else np.ones(0, bool)
    if EXACT and not (same.all() and same_bits(g["t"][k], o["t"][k]) and (g["front"][k] == o["front"][k]).all() and
                      same_bits(g["N"][k], o["N"][k]) and same_bits(g["p"][k], o["p"][k])):
        msgs.append(f"closest side {side} lights {int(wl)}: not bit-identical (same {same.mean():.5f}, t {np.mean(g['t'][k] == o['t'][k]):.5f}, "
                    f"N {np.mean((g['N'][k] == o['N'][k]).all(-1)):.5f}, p {np.mean((g['p'][k] == o['p'][k]).all(-1)):.5f})")
    if same.mean() < 0.995 or (k.any() and (ok_t.mean() < 0.999 or ok_f.mean() < 0.999 or ok_n.mean() < 0.995)):
        msgs.append(f"closest side {side} lights {int(wl)}: same {same.mean():.4f} t {ok_t.mean():.4f} front {ok_f.mean():.4f} N {ok_n.mean():.4f}")
    if os.environ.get("FUZZ_VERBOSE") and not same.all():
        for i in np.nonzero(~same)[0][:8]:
            print(f"      ray {i}: gpu obj {g['obj'][i]} face {g['face'][i]} mtl {g['mtl_id'][i]} light {g['light'][i]} t {g['t'][i]:.7f} front {g['front'][i]} | "
                  f"oracle obj {o['obj'][i]} face {o['face'][i]} mtl {o['mtl_id'][i]} light {o['light'][i]} t {o['t'][i]:.7f} front {o['front'][i]}")
    # ShadeInfoAdapter::TraceSecondaryRay of the legacy mode from a few of those hit points
    import ctypes as C
    hits = np.nonzero(k & (o["obj"] >= 0))[0][:12]
    worst_sec = 1.0
    for i in hits:
        at = _ffi.ShadePoint()
        at.P = (C.c_float * 3)(*[float(x) for x in o["p"][i]])
        at.N = (C.c_float * 3)(*[float(x) for x in o["N"][i]])
        at.obj, at.bounce, at.diffuse_depth = int(o["obj"][i]), int(rng.integers(0, 4)), int(rng.integers(0, 2))
        at.pixel, at.sample, at.node = int(rng.integers(0, 1000)), int(rng.integers(0, 8)), int(rng.integers(0, 40))
        dirs = rng.normal(size=(16, 3)).astype(np.float32)
        dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        child, rec = int(rng.integers(1, 4)), bool(rng.random() < 0.8)
        gsec = gs.trace_secondary(at, dirs, recursive=rec, child=child, seed=seed + 5)
        osec = osc.trace_secondary(at, dirs, recursive=rec, child=child, seed=seed + 5)
        okc = (np.abs(gsec[:, :3] - osec[:, :3]).max(-1) <= 2e-3 * (1 + np.abs(osec[:, :3]).max(-1))).mean()
        okd = (np.abs(gsec[:, 3] - osec[:, 3]) <= 1e-4 * (1 + np.abs(osec[:, 3]))).mean()
        worst_sec = min(worst_sec, okc, okd)
    if worst_sec < 0.8:
        msgs.append(f"trace_secondary: worst call agrees on {worst_sec:.3f} of its 16 rays")
    seg = rand_rays(rng, 3000, segments=True)
    nrm = rng.normal(size=(3000, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    ut = bool(rng.random() < 0.7)
    gv, ov = gs.trace_visibility(seg, nrm, use_transmittance=ut), osc.trace_visibility(seg, nrm, use_transmittance=ut)
    okv = np.abs(gv - ov) <= 2e-3
    if EXACT and not same_bits(gv, ov):
        msgs.append(f"visibility transmittance {int(ut)}: {np.mean(gv == ov):.5f} bit-identical")
    if okv.mean() < 0.995:
        msgs.append(f"visibility transmittance {int(ut)}: {okv.mean():.4f}")
    ft = int(rng.integers(0, 4))
    rp = render_params(spp_min=int(rng.integers(1, 4)), spp_max=int(rng.integers(4, 10)), ci_halfwidth=float(rng.choice([0.005, 0.05])),
                       max_bounces=int(rng.choice([2, 6])), filter_type=ft, filter_radius=float(rng.choice([0.0, 0.7, 1.0, 2.5])),
                       tile_size=int(rng.choice([4, 16])), seed=seed + 17)
    gi, oi = gs.render(cam, rp), osc.render(cam, rp)
    eq = (gi["samples"] == oi["samples"]).mean()
    dl = np.abs(gi["linear"] - oi["linear"]).mean() / (np.abs(oi["linear"]).mean() + 1e-3)
    dz = (np.abs(gi["z"] - oi["z"]) <= 1e-4 * (1 + np.abs(oi["z"]))).mean()   # transformed meshes: t to 1e-4
    if os.environ.get("FUZZ_VERBOSE") and (eq < 1.0 or dz < 1.0):
        ys, xs = np.nonzero((gi["samples"] != oi["samples"]) | (np.abs(gi["z"] - oi["z"]) > 1e-5 * (1 + np.abs(oi["z"]))))
        for y, x in list(zip(ys, xs))[:10]:
            print(f"      pixel ({x},{y}): samples {gi['samples'][y, x]}/{oi['samples'][y, x]} z {gi['z'][y, x]:.6f}/{oi['z'][y, x]:.6f} "
                  f"linear {gi['linear'][y, x]} / {oi['linear'][y, x]}")
    if EXACT and (eq < 1.0 or not same_bits(gi["z"], oi["z"])):
        msgs.append(f"adaptive render: sample maps equal {eq:.5f}, z bit-identical {np.mean(gi['z'] == oi['z']):.5f}")
    if eq < 0.97 or dl > 0.03 or dz < 0.99 or not np.isfinite(gi["linear"]).all():
        msgs.append(f"adaptive render filter {ft}/{rp.filter_radius:.1f} spp {rp.spp_min}..{rp.spp_max}: samples equal {eq:.4f} linear diff {dl:.4f} z {dz:.4f}")
    return msgs


EXACT = False


def same_bits(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and bool(((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all())


def agree(g, o, tol):
    d = np.abs(g[..., :3] - o[..., :3]).max(-1)
    return float((d <= tol * (1 + np.abs(o[..., :3]).max(-1))).mean())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="0:40")
    ap.add_argument("--mode", default="both", choices=["path", "legacy", "both"])
    ap.add_argument("--photons", action="store_true")
    ap.add_argument("--big", action="store_true", help="images of 160x120 and more: paths beyond the tail kernel's threshold, several waves per queue segment")
    ap.add_argument("--spice", action="store_true", help="second-stage mutations: index arrays, degenerate / duplicate faces, extreme materials")
    ap.add_argument("--api", action="store_true", help="also trace_closest / trace_visibility on random rays and an adaptive render")
    ap.add_argument("--spp", type=int, default=4)
    ap.add_argument("--tol", type=float, default=2e-3)
    ap.add_argument("--min-agree", type=float, default=0.98)
    ap.add_argument("--exact", action="store_true", help="round-2 bar: path-mode samples, primary distances, closest hits, visibility, photon "
                    "counts and estimates bit-identical; legacy mode within 1e-5 on 99.99 %% of samples")
    a = ap.parse_args()
    global EXACT
    EXACT = a.exact
    lo, hi = (int(x) for x in a.seeds.split(":"))
    modes = {"path": [_ffi.SHADING_PATH], "legacy": [_ffi.SHADING_LEGACY], "both": [_ffi.SHADING_PATH, _ffi.SHADING_LEGACY]}[a.mode]
    worst, bad = 1.0, []
    for seed in range(lo, hi):
        sd, cam, rng = rand_scene(seed)
        if a.spice:
            spice_scene(sd, cam, seed)
        if a.big:
            cam.width, cam.height = 160 + 8 * (seed % 7), 120 + 4 * (seed % 5)
        gs, osc = rtx.Scene(sd), OracleScene(sd)
        gp = op = None
        if a.photons and sd.lights:
            gp = rtx.Photons(gs, num_photons=20000, seed=seed + 1)
            op = OraclePhotons(osc, num_photons=20000, seed=seed + 1)
        if a.api:
            msgs = api_leg(gs, osc, sd, cam, rng, seed)
            for m in msgs:
                print(f"seed {seed:4d} api: {m}   <-- CHECK", flush=True)
            if msgs:
                bad.append((seed, "api"))
            else:
                print(f"seed {seed:4d} api ok", flush=True)
        if gp is not None:
            # photon maps: stored counts, and the density estimate at the oracle's own photon positions
            gi, oi = gp.info(), op.info()
            msgs = []
            for k in ("n_global", "n_caustic"):
                if abs(gi[k] - oi[k]) > (0 if EXACT else max(3, 0.005 * oi[k])):
                    msgs.append(f"{k} {gi[k]} / {oi[k]}")
            if abs(gi["photon_radius"] - oi["photon_radius"]) > 1e-6 or abs(gi["caustic_radius"] - oi["caustic_radius"]) > 1e-6:
                msgs.append(f"radii {gi['photon_radius']},{gi['caustic_radius']} / {oi['photon_radius']},{oi['caustic_radius']}")
            for which in (0, 1):
                ph = op.download(which)
                if len(ph) < 10:
                    continue
                sel = ph[:: max(1, len(ph) // 200)]
                nn = -sel["dir"] / np.maximum(1e-9, np.linalg.norm(sel["dir"], axis=1, keepdims=True))
                ge, oe = gp.estimate(which, sel["pos"], nn), op.estimate(which, sel["pos"], nn)
                sc = np.abs(oe).max() + 1e-12
                okp = (np.abs(ge - oe).max(-1) <= 0.03 * sc).mean()
                if EXACT and not same_bits(ge, oe):
                    msgs.append(f"estimate map {which}: {np.mean((ge == oe).all(-1)):.4f} of the probes bit-identical")
                if okp < 0.95:
                    msgs.append(f"estimate map {which}: {okp:.3f} within 3 % of the largest ({len(sel)} probes of {len(ph)} photons)")
            for m in msgs:
                print(f"seed {seed:4d} photons: {m}   <-- CHECK", flush=True)
            if msgs:
                bad.append((seed, "photons"))
            else:
                print(f"seed {seed:4d} photons ok: global {gi['n_global']} caustic {gi['n_caustic']}", flush=True)
        for mode in modes:
            rp = render_params(spp_min=a.spp, max_bounces=int(rng.choice([1, 3, 8, 64])), seed=seed * 7919 + 1, shading_mode=mode,
                               use_photons=int(gp is not None))
            g, st = gs.render_samples(cam, rp, gp)
            o = osc.render_samples(cam, rp, op)
            # primary hit distance: exact or both misses
            dz = np.abs(g[..., 3] - o[..., 3]) <= 1e-4 * (1 + np.abs(o[..., 3]))
            ag = agree(g, o, a.tol)
            fin = bool(np.isfinite(g).all())
            ofin = bool(np.isfinite(o).all())
            worst = min(worst, ag)
            # two ill-conditioned constructions of the reference itself: the legacy mode biases its secondary
            # rays along the bump-mapped normal (about 1 in 1000 then starts within rounding of its own surface, and a
            # recursion tree holds hundreds of them); an emissive background maps escaping directions through the
            # camera's screen projection, which is singular on the plane perpendicular to the view axis
            need = a.min_agree
            if mode == _ffi.SHADING_LEGACY and any(m.bump_tex >= 0 for m in sd.materials):
                need = min(need, 0.80)
            if sd.background_emission > 0.0:
                need = min(need, 0.95)
            flag = "" if (ag >= need and dz.mean() >= 0.999 and (fin or not ofin)) else "   <-- CHECK"
            ex = float(((g.view(np.uint32) == o.view(np.uint32)) | (np.isnan(g) & np.isnan(o))).all(-1).mean())
            if EXACT:
                ok = ex == 1.0   # both shading modes: every sample the oracle's bits
                flag = "" if ok else "   <-- CHECK"
            if flag:
                bad.append((seed, mode))
            print(f"seed {seed:4d} mode {mode} {cam.width:3d}x{cam.height:<3d} objs {len(sd.meshes):2d} tris {sd.n_triangles:4d} lights {len(sd.lights)} "
                  f"bounces {rp.max_bounces:2d}  agree {ag:.4f}  exact {ex:.5f}  z {dz.mean():.4f}  finite gpu/oracle {int(fin)}/{int(ofin)}  "
                  f"mean {g[..., :3].mean():.4f}/{o[..., :3].mean():.4f}{flag}", flush=True)
    print(f"worst agreement {worst:.4f}; scenes to check: {bad}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
