"""Where the flattened product, the oracle's flattened walk and its level-by-level walk of tests/node_tree.py differ (GPU box)."""
import numpy as np
import raytracer_amd as ra
from raytracer_amd import _ffi
from tests import node_tree
from tests.helpers import random_rays, global_prim
from tests.oracle_ffi import OracleScene

sd, parent, tms, node_mesh = node_tree.build()
gs = ra.Scene(sd)
flat = OracleScene(sd)
tree = OracleScene(sd)
tree.set_tree(parent, tms, node_mesh)
rays = random_rays(np.random.default_rng(3), 200000, tree.bounds())
for side in (_ffi.HIT_FRONT_AND_BACK, _ffi.HIT_FRONT):
    g = gs.trace_closest(rays, side, with_lights=True)
    f = flat.trace_closest(rays, side, with_lights=True)
    t = tree.trace_closest(rays, side, with_lights=True)
    for name, a, b in (("gpu-flat", g, f), ("gpu-tree", g, t), ("flat-tree", f, t)):
        gp, op = global_prim(a, gs.face_base), global_prim(b, gs.face_base)
        same = (gp == op) & (a["light"] == b["light"])
        hit = same & ((gp >= 0) | (a["light"] >= 0))
        rel = np.abs(a["t"][hit] - b["t"][hit]) / np.maximum(np.abs(b["t"][hit]), 1e-20)
        print(side, name, "id mismatches", int((~same).sum()), "bit-identical t", float((a["t"][hit] == b["t"][hit]).mean()), "max rel", rel.max(),
              "per object max", {int(o): float(rel[a["obj"][hit] == o].max()) for o in np.unique(a["obj"][hit])})
