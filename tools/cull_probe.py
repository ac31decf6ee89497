#!/usr/bin/env python3
"""How many camera samples of the full-size C3 scene change when the kernels repeat the reference's root and object box
tests (general kernels, RTX_AMD_GENERAL_KERNELS=1) - the tests the lean kernels for world-space triangle scenes leave out
(DESIGN.md 4 item 8). Renders per-sample radiance twice and counts samples whose bits differ."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra
from raytracer_amd import synth
from raytracer_amd.scene import render_params

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 2
sd = synth.build_scene(1.0)
cam = synth.cerealbar_camera(1920, 1080)
rp = render_params(spp_min=spp, max_bounces=64)
os.environ.pop("RTX_AMD_GENERAL_KERNELS", None)
lean, st0 = ra.Scene(sd).render_samples(cam, rp)
os.environ["RTX_AMD_GENERAL_KERNELS"] = "1"
gen, st1 = ra.Scene(sd).render_samples(cam, rp)
diff = (lean.view(np.uint32) != gen.view(np.uint32)).any(-1)
rays0 = st0["rays_primary"] + st0["rays_bounce"] + st0["rays_shadow"]
rays1 = st1["rays_primary"] + st1["rays_bounce"] + st1["rays_shadow"]
print(f"samples {diff.size}, differing {int(diff.sum())} ({diff.mean():.3e}); ray segments lean {rays0} general {rays1}; "
      f"frame ms lean {st0['total_ms']:.1f} general {st1['total_ms']:.1f}")
if diff.any():
    d = np.abs(lean[diff][:, :3] - gen[diff][:, :3]).max(-1)
    print("largest per-sample difference", float(d.max()), "median", float(np.median(d)))
