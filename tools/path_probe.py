"""Development probe: GPU path tracer vs oracle on a small cerealbar-synth, prints parity metrics."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as ra
from raytracer_amd import synth, _ffi
from raytracer_amd.scene import render_params
from tests.oracle_ffi import OracleScene, OraclePhotons

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.003
W, H, spp = 96, 54, 8
sd = synth.build_scene(scale)
cam = synth.cerealbar_camera(W, H)
gs, osc = ra.Scene(sd), OracleScene(sd)
for mb in (1, 4, 64):
    rp = render_params(spp_min=spp, max_bounces=mb)
    t = time.time(); g, st = gs.render_samples(cam, rp); tg = time.time() - t
    t = time.time(); o = osc.render_samples(cam, rp); to = time.time() - t
    d = np.abs(g[..., :3] - o[..., :3]).max(-1)
    tol = 1e-3 * (1 + np.abs(o[..., :3]).max(-1))
    pdm = np.abs(g[..., 3] - o[..., 3]) <= 1e-4 * np.abs(o[..., 3])
    print(f"max_bounces={mb}: gpu {tg:.2f}s oracle {to:.2f}s  samples within tol {np.mean(d <= tol):.5f}  primaryDist match {pdm.mean():.5f} "
          f"mean|d| {d.mean():.3e} mean L {o[..., :3].mean():.4f}  rays {st['rays_primary']},{st['rays_bounce']},{st['rays_shadow']}", flush=True)
    bad = np.argwhere(d > tol)
    for b in bad[:6]:
        print("   bad", tuple(b), g[tuple(b)], o[tuple(b)])
rp = render_params(spp_min=spp, max_bounces=64)
gi = gs.render(cam, rp); oi = osc.render(cam, rp)
rmse = np.sqrt(np.mean((np.minimum(gi["linear"], 10) - np.minimum(oi["linear"], 10)) ** 2))
print("image rmse", rmse, "mean", oi["linear"].mean(), "rel", rmse / oi["linear"].mean(), "z match", np.mean(np.abs(gi["z"] - oi["z"]) <= 1e-4 * np.abs(oi["z"])),
      "samples eq", (gi["samples"] == oi["samples"]).all(), "rgb8 max diff", np.abs(gi["rgb8"].astype(int) - oi["rgb8"].astype(int)).max(), flush=True)
print("oracle rays", oi["rays"], "gpu", {k: gi["stats"][k] for k in ("rays_primary", "rays_bounce", "rays_shadow")})
# adaptive
rp = render_params(spp_min=8, spp_max=32, max_bounces=8, ci_halfwidth=0.02)
ga = gs.render(cam, rp); oa = osc.render(cam, rp)
print("adaptive: sample maps equal", np.mean(ga["samples"] == oa["samples"]), "hist gpu", np.bincount(ga["samples"].ravel() // 8), "oracle", np.bincount(oa["samples"].ravel() // 8))
# visibility
from tests.helpers import random_rays
rng = np.random.default_rng(5)
rays = random_rays(rng, 20000, osc.bounds(), tmax=3.0)
nrm = rng.normal(size=(20000, 3)).astype(np.float32)
for ut in (True, False):
    gv = gs.trace_visibility(rays, nrm, ut); ov = osc.trace_visibility(rays, nrm, ut)
    print("visibility", ut, "max diff", np.abs(gv - ov).max(), "frac equal", np.mean(gv == ov), "mean", ov.mean())
# photons
t = time.time(); gp = ra.Photons(gs, num_photons=200000); print("gpu photons", gp.info(), gp.stats["rays_photon"], time.time() - t)
t = time.time(); op = OraclePhotons(osc, 200000); print("oracle photons", op.info(), time.time() - t)
for which in (0, 1):
    a, b = gp.download(which), op.download(which)
    print("which", which, len(a), len(b))
    if len(a) == len(b) and len(a):
        ka = np.lexsort((a["pos"][:, 2], a["pos"][:, 1], a["pos"][:, 0])); kb = np.lexsort((b["pos"][:, 2], b["pos"][:, 1], b["pos"][:, 0]))
        print("  pos max diff", np.abs(a["pos"][ka] - b["pos"][kb]).max(), "power rel", (np.abs(a["power"][ka] - b["power"][kb]) / (1e-9 + np.abs(b["power"][kb]))).max())
probe = b["pos"][:2000] if len(b) else np.zeros((1, 3), np.float32)
nn = np.tile(np.array([[0, 1, 0]], np.float32), (len(probe), 1))
for which in (0, 1):
    ge, oe = gp.estimate(which, probe, nn), op.estimate(which, probe, nn)
    print("estimate", which, "max rel diff", (np.abs(ge - oe).max() / (1e-12 + np.abs(oe).max())), "mean", oe.mean())
rp = render_params(spp_min=4, max_bounces=4, use_photons=1)
g, _ = gs.render_samples(cam, rp, gp); o = osc.render_samples(cam, rp, op)
d = np.abs(g[..., :3] - o[..., :3]).max(-1); tol = 1e-3 * (1 + np.abs(o[..., :3]).max(-1))
print("with caustics: samples within tol", np.mean(d <= tol))
