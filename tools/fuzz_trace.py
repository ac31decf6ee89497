#!/usr/bin/env python3
"""tools/fuzz_trace.py SEED [MODE]: for the samples of a fuzz scene whose radiance differs between the HIP path and
the oracle, the radiance at max_bounces = 1, 2, 3, 4 side by side (the first limit at which they part tells which
vertex of the path differs). [FUZZ_SPICE=1] as in fuzz_diag.py. TEST INFRASTRUCTURE (uses the oracle)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rtx  # noqa: E402
from raytracer_amd import _ffi  # noqa: E402
from raytracer_amd.scene import render_params  # noqa: E402
from tests.oracle_ffi import OracleScene  # noqa: E402
from tools.fuzz_parity import rand_scene, spice_scene  # noqa: E402

seed = int(sys.argv[1])
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
sd, cam, rng = rand_scene(seed)
if os.environ.get("FUZZ_SPICE"):
    spice_scene(sd, cam, seed)
if os.environ.get("FUZZ_NO_BUMP"):
    for m in sd.materials:
        m.bump_tex = -1
if os.environ.get("FUZZ_NO_ENVTEX"):
    sd.environment_tex = -1
if os.environ.get("FUZZ_NO_DOF"):
    cam.dof = 0.0
if os.environ.get("FUZZ_NO_TEX"):
    for m in sd.materials:
        m.diffuse_tex = m.specular_tex = m.reflection_tex = m.refraction_tex = m.emission_tex = -1
gs, osc = rtx.Scene(sd), OracleScene(sd)
prim, t, front = gs.render_primary(cam, hit_side=_ffi.HIT_FRONT_AND_BACK)
G, O = {}, {}
for nb in (1, 2, 3, 4):
    rp = render_params(spp_min=4, max_bounces=nb, seed=seed * 7919 + 1, shading_mode=mode)
    G[nb], _ = gs.render_samples(cam, rp)
    O[nb] = osc.render_samples(cam, rp)
bad = {nb: np.abs(G[nb][..., :3] - O[nb][..., :3]).max(-1) > 2e-3 * (1 + np.abs(O[nb][..., :3]).max(-1)) for nb in G}
first = np.zeros(bad[1].shape, int)
for nb in (4, 3, 2, 1):
    first[bad[nb]] = nb
for nb in (1, 2, 3, 4):
    print(f"first differ at max_bounces {nb}: {(first == nb).sum()} samples")
shown = 0
for y, x, s in zip(*np.nonzero(first > 0)):
    if shown >= int(os.environ.get("FUZZ_SHOW", "12")):
        break
    shown += 1
    print(f"sample ({x},{y},{s}) primary face {prim[y, x]} front {front[y, x]} t {t[y, x]:.5f}")
    for nb in (1, 2, 3, 4):
        print(f"    max_bounces {nb}: gpu {G[nb][y, x, s, :3]}  oracle {O[nb][y, x, s, :3]}{'   <--' if bad[nb][y, x, s] else ''}")
