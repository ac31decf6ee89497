"""Measures how close per-sample radiance is to the oracle at several tolerances (what the parity bars are set from).
  python tools/parity_levels.py            (GPU box; prints one line per configuration)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra  # noqa: E402
from raytracer_amd import synth  # noqa: E402
from raytracer_amd.scene import render_params  # noqa: E402
from tests.oracle_ffi import OracleScene  # noqa: E402


def levels(g, o):
    d = np.abs(g[..., :3] - o[..., :3]).max(-1)
    ref = 1 + np.abs(o[..., :3]).max(-1)
    return " ".join(f"{tol:g}:{float((d <= tol * ref).mean()):.5f}" for tol in (1e-3, 1e-4, 1e-5, 1e-6)) + f" exact:{float((d == 0).mean()):.5f}"


def main():
    ra.set_device(0)
    for scale, W, H, spp in ((0.003, 96, 54, 8), (0.002, 64, 36, 4), (0.02, 160, 90, 4)):
        sd = synth.build_scene(scale)
        cam = synth.cerealbar_camera(W, H)
        gs, osc = ra.Scene(sd), OracleScene(sd)
        for mb in (1, 4, 8, 64):
            rp = render_params(spp_min=spp, max_bounces=mb)
            g, _ = gs.render_samples(cam, rp)
            o = osc.render_samples(cam, rp)
            print(f"scale {scale} {W}x{H} spp {spp} max_bounces {mb:2d}: {levels(g, o)}", flush=True)
        os.environ["RTX_AMD_TAIL_PATHS"] = "0"
        rp = render_params(spp_min=spp, max_bounces=64)
        a, _ = gs.render_samples(cam, rp)
        del os.environ["RTX_AMD_TAIL_PATHS"]
        b, _ = gs.render_samples(cam, rp)
        print(f"   tail kernel vs shade kernel: {levels(a, b)}", flush=True)


if __name__ == "__main__":
    main()
