#!/usr/bin/env python3
"""Where does a fuzz scene disagree? [FUZZ_SPICE=1] tools/fuzz_diag.py SEED [MODE] [BOUNCES]: per primary-hit object, the fraction of
samples whose radiance differs between the HIP path and the oracle, with that object's kind, transform and material.
TEST INFRASTRUCTURE (uses the oracle)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rtx  # noqa: E402
from raytracer_amd import _ffi  # noqa: E402
from raytracer_amd.scene import render_params  # noqa: E402
from tests.oracle_ffi import OracleScene  # noqa: E402
from tools.fuzz_parity import rand_scene, spice_scene  # noqa: E402


def main():
    seed = int(sys.argv[1])
    mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    sd, cam, rng = rand_scene(seed)
    if os.environ.get("FUZZ_SPICE"):
        spice_scene(sd, cam, seed)
    gs, osc = rtx.Scene(sd), OracleScene(sd)
    prim, t, front = gs.render_primary(cam, hit_side=_ffi.HIT_FRONT_AND_BACK)
    fb = sd.face_base()
    obj_of = np.full(prim.shape, -1)
    for m in range(len(sd.meshes)):
        lo = fb[m]
        hi = lo + sd.meshes[m].nf
        obj_of[(prim >= lo) & (prim < hi)] = m
    for nb in ([int(sys.argv[3])] if len(sys.argv) > 3 else [1, 2, 3, 8]):
        rp = render_params(spp_min=4, max_bounces=nb, seed=seed * 7919 + 1, shading_mode=mode)
        g, _ = gs.render_samples(cam, rp)
        o = osc.render_samples(cam, rp)
        d = np.abs(g[..., :3] - o[..., :3]).max(-1)
        badm = d > 2e-3 * (1 + np.abs(o[..., :3]).max(-1))
        print(f"== seed {seed} mode {mode} max_bounces {nb}: disagree {badm.mean():.4f}")
        for m in sorted(set(obj_of.ravel().tolist())):
            sel = obj_of == m
            n = int(sel.sum())
            bm = badm[sel]
            if m < 0:
                print(f"   miss: pixels {n:4d} bad {bm.mean():.3f}")
                continue
            me = sd.meshes[m]
            mt = sd.materials[me.material] if me.material >= 0 else None
            desc = "none" if mt is None else (f"type {mt.type} kd {np.round(mt.diffuse, 2)} ks {np.round(mt.specular, 2)} refl {np.round(mt.reflection, 2)} "
                                             f"refr {np.round(mt.refraction, 2)} ke {np.round(mt.emission, 2)} gloss {mt.glossiness:.0f} ior {mt.ior:.2f} "
                                             f"tex d{mt.diffuse_tex} s{mt.specular_tex} r{mt.reflection_tex} t{mt.refraction_tex} e{mt.emission_tex} b{mt.bump_tex}")
            det = float(np.linalg.det(me.xform.reshape(3, 4)[:, :3]))
            print(f"   obj {m} kind {me.kind} nf {me.nf} vn {me.vn is not None} vt {me.vt is not None} det {det:+.3f} mtl {me.material}: "
                  f"pixels {n:4d} bad {bm.mean():.3f} front {front[sel].mean():.2f} | {desc}")
            if bm.mean() > 0.05:
                ys, xs = np.nonzero(sel)
                k = 0
                for y, x in zip(ys, xs):
                    for s in range(4):
                        if badm[y, x, s] and k < 3:
                            print(f"        ({x},{y},{s}) gpu {g[y, x, s]} oracle {o[y, x, s]}")
                            k += 1
    print(f"lights: {[(np.round(l.position, 2).tolist(), l.size, l.attenuation) for l in sd.lights]} env {sd.environment} envtex {sd.environment_tex} "
          f"bg {sd.background} bgtex {sd.background_tex} dof {cam.dof:.3f}")
    for i, tmap in enumerate(sd.texmaps):
        print(f"tex {i}: {tmap.texels.shape} {tmap.texels.dtype} xform identity {bool(np.allclose(tmap.xform, [1,0,0,0,0,1,0,0,0,0,1,0]))}")


if __name__ == "__main__":
    main()
