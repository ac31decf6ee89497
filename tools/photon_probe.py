"""Development probe (not the judged bench): config C5 timing — photon map build with 10^7 emissions on cerealbar-synth
and a path-traced frame that gathers from the caustic map."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as ra
from raytracer_amd import synth
from raytracer_amd.scene import render_params

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
nph = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10_000_000
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 8
ra.set_device(0)
sc = ra.Scene(synth.build_scene(scale))
for it in range(2):
    t0 = time.time()
    ph = ra.Photons(sc, num_photons=nph, max_bounces=2)
    dt = time.time() - t0
    info = ph.info()
    print(f"photon build {nph} emitted: {dt*1e3:.1f} ms  {info}", flush=True)
    if it == 0:
        ph.close()
cam = synth.cerealbar_camera()
for use in (0, 1):
    rp = render_params(spp_min=spp, max_bounces=64, use_photons=use)
    sc.render(cam, rp, photons=ph if use else None)
    t0 = time.time()
    r = sc.render(cam, rp, photons=ph if use else None)
    dt = time.time() - t0
    st = r["stats"]
    print(f"frame {spp} spp use_photons={use}: {dt*1e3:.1f} ms  extend {st['extend_ms']:.1f} shadow {st['shadow_ms']:.1f} shade {st['shade_ms']:.1f}  mean {r['linear'].mean():.5f}", flush=True)
