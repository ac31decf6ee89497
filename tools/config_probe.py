"""Development probe (not the judged bench): wall time of BASELINE.json's configs C2 and C4 on cerealbar-synth."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as ra
from raytracer_amd import synth
from raytracer_amd.scene import render_params

ra.set_device(0)
sc = ra.Scene(synth.build_scene(1.0))
cam = synth.cerealbar_camera()
for name, rp in (("C2 16 spp, primary + soft shadows", render_params(spp_min=16, max_bounces=1)),
                 ("C3 128 spp path traced", render_params(spp_min=128, max_bounces=64)),
                 ("C4 adaptive 128..1024 spp", render_params(spp_min=128, spp_max=1024, max_bounces=64, ci_halfwidth=0.005))):
    sc.render(cam, render_params(spp_min=2, max_bounces=2))
    t0 = time.time()
    r = sc.render(cam, rp)
    dt = time.time() - t0
    st = r["stats"]
    rays = st["rays_primary"] + st["rays_bounce"] + st["rays_shadow"]
    s = r["samples"]
    print(f"{name}: {dt:.3f} s  {rays/dt/1e6:.0f} Mrays/s  rays {rays/1e9:.2f} G  samples/pixel mean {s.mean():.1f} max {s.max()}  "
          f"extend {st['extend_ms']:.0f} shadow {st['shadow_ms']:.0f} shade {st['shade_ms']:.0f} ms", flush=True)
