"""How many per-sample results are bit-identical to the oracle, test by test (GPU box): the numbers the bars in tests/ are
set from. Prints one line per configuration: share bit-identical, share within 1e-6 * (1 + |ref|)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra  # noqa: E402
from raytracer_amd import synth, _ffi  # noqa: E402
from raytracer_amd.scene import render_params  # noqa: E402
from tests.oracle_ffi import OracleScene, OraclePhotons  # noqa: E402


def line(tag, g, o):
    d = np.abs(g[..., :3] - o[..., :3]).max(-1)
    ex = (g[..., :3].view(np.uint32) == o[..., :3].view(np.uint32)).all(-1)
    print(f"{tag:60s} exact {ex.mean():.6f}  1e-6 {(d <= 1e-6 * (1 + np.abs(o[..., :3]).max(-1))).mean():.6f}  dist exact {(g[..., 3] == o[..., 3]).mean():.6f}", flush=True)


ra.set_device(0)
sd = synth.build_scene(0.003)
cam = synth.cerealbar_camera(96, 54)
gs, osc = ra.Scene(sd), OracleScene(sd)
for mode, name in ((_ffi.SHADING_PATH, "path"), (_ffi.SHADING_LEGACY, "legacy")):
    for mb in (1, 4, 64):
        rp = render_params(spp_min=8, max_bounces=mb, shading_mode=mode)
        line(f"cerealbar 0.003 {name} max_bounces {mb}", gs.render_samples(cam, rp)[0], osc.render_samples(cam, rp))
gp, op = ra.Photons(gs, num_photons=300000), OraclePhotons(osc, 300000)
for mode, name in ((_ffi.SHADING_PATH, "path"), (_ffi.SHADING_LEGACY, "legacy")):
    rp = render_params(spp_min=4, max_bounces=4, use_photons=1, shading_mode=mode)
    c64 = synth.cerealbar_camera(64, 36)
    line(f"cerealbar 0.003 {name} + photons", gs.render_samples(c64, rp, gp)[0], osc.render_samples(c64, rp, op))
from tests import test_stress_gpu as ts  # noqa: E402
for nl in (0, 1, 3, 8):
    s2 = ts.room(n_lights=nl)
    g2, o2 = ra.Scene(s2), OracleScene(s2)
    c2 = ts.cam_for(1.0, 64, 48)
    for mode, name in ((_ffi.SHADING_PATH, "path"), (_ffi.SHADING_LEGACY, "legacy")):
        rp = render_params(spp_min=4, max_bounces=16, shading_mode=mode)
        line(f"room lights {nl} {name}", g2.render_samples(c2, rp)[0], o2.render_samples(c2, rp))
