"""Where do full-size path samples part from the oracle? For a random pixel subset: share of samples within tolerance
as a function of max_bounces, and the size of the differences (GPU box)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra  # noqa: E402
from raytracer_amd import synth  # noqa: E402
from raytracer_amd.scene import render_params  # noqa: E402
from tests.oracle_ffi import OracleScene  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
ra.set_device(0)
sd = synth.build_scene(scale)
cam = synth.cerealbar_camera()
gs, osc = ra.Scene(sd), OracleScene(sd)
spp, k = 2, 4096
rng = np.random.default_rng(78)
px, py = rng.integers(0, cam.width, k), rng.integers(0, cam.height, k)
xys = np.stack([np.repeat(px, spp), np.repeat(py, spp), np.tile(np.arange(spp), k)], 1).astype(np.int32)
prev_bad = None
for mb in (1, 2, 3, 4, 6, 8, 16, 64):
    rp = render_params(spp_min=spp, max_bounces=mb)
    g, _ = gs.render_samples(cam, rp)
    o = osc.trace_samples(cam, rp, xys)
    gsub = g[xys[:, 1], xys[:, 0], xys[:, 2]]
    d = np.abs(gsub[:, :3] - o[:, :3]).max(-1)
    ref = 1 + np.abs(o[:, :3]).max(-1)
    bad = d > 1e-4 * ref
    new = bad if prev_bad is None else (bad & ~prev_bad)
    print(f"max_bounces {mb:2d}: within 1e-4 {1 - bad.mean():.5f}  1e-3 {(d <= 1e-3 * ref).mean():.5f}  1e-2 {(d <= 1e-2 * ref).mean():.5f}  "
          f"newly bad {int(new.sum())}  median |d| of bad {np.median(d[bad]) if bad.any() else 0:.4f}  mean signed diff {float((gsub[:, :3] - o[:, :3]).mean()):+.2e}", flush=True)
    if mb == 1 and bad.any():
        idx = np.nonzero(bad)[0][:12]
        for i in idx:
            print("   ", xys[i], gsub[i], o[i])
    prev_bad = bad
