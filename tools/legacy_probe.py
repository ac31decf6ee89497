"""Parity numbers of the legacy shading mode on the GPU box (per-sample agreement with the oracle) for DESIGN.md."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as ra
from raytracer_amd import synth, _ffi
from raytracer_amd.scene import render_params
from tests.oracle_ffi import OracleScene

sd = synth.build_scene(0.003)
cam = synth.cerealbar_camera(96, 54)
gs, osc = ra.Scene(sd), OracleScene(sd)
rp = render_params(spp_min=8, shading_mode=_ffi.SHADING_LEGACY)
g, st = gs.render_samples(cam, rp)
o = osc.render_samples(cam, rp)
d = np.abs(g[..., :3] - o[..., :3]).max(-1)
ref = np.abs(o[..., :3]).max(-1)
for tol in (1e-5, 1e-4, 1e-3, 1e-2):
    print(f"within {tol:g}*(1+|ref|): {(d <= tol * (1 + ref)).mean() * 100:.3f} %")
print("max |radiance|", float(o[..., :3].max()), "mean", float(o[..., :3].mean()), "gpu mean", float(g[..., :3].mean()))
print("rays per camera sample: secondary", st["rays_bounce"] / st["rays_primary"], "shadow", st["rays_shadow"] / st["rays_primary"])
