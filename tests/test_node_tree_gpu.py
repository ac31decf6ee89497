"""The flattened product against the reference's walk of a real node tree three levels deep (IntersectNodeAdapter.h:53-95,
restated level by level in the oracle): closest hits and shadow queries at north_star's bar - ids >= 99.99 %, t <= 1e-5 relative on triangle meshes (analytic objects:
node_tree.compare_tree).
(Against the oracle's own flattened walk the product is bit-identical: tests/test_closest_hit_gpu.py.)"""
import numpy as np
import pytest

import raytracer_amd as ra
from raytracer_amd import _ffi
from tests import node_tree
from tests.helpers import random_rays
from tests.oracle_ffi import OracleScene

pytestmark = pytest.mark.gpu


def test_flattened_product_against_the_tree_walk():
    sd, parent, tms, node_mesh = node_tree.build()
    gs = ra.Scene(sd)
    tree = OracleScene(sd)
    tree.set_tree(parent, tms, node_mesh)
    rays = random_rays(np.random.default_rng(3), 40000, tree.bounds())
    for side in (_ffi.HIT_FRONT_AND_BACK, _ffi.HIT_FRONT):
        g = gs.trace_closest(rays, side, with_lights=True)
        o = tree.trace_closest(rays, side, with_lights=True)
        same, hit = node_tree.compare_tree(g, o, gs.face_base, node_tree.analytic_objects(sd))
        assert hit.sum() > 5000
        assert set(np.unique(g["obj"][hit])) >= set(range(len(sd.meshes)))
    # Visible() along segments between surface points and the light
    o = tree.trace_closest(rays, _ffi.HIT_FRONT_AND_BACK)
    hit = o["obj"] >= 0
    P, N = o["p"][hit][:8000], o["N"][hit][:8000]
    L = np.array(sd.lights[0].position, np.float32)
    d = L[None] - P
    dist = np.linalg.norm(d, axis=1).astype(np.float32)
    ray_dt, _, _ = _ffi.np_dtypes()
    seg = np.zeros(P.shape[0], ray_dt)
    seg["o"], seg["d"], seg["tmax"] = P, d / dist[:, None], dist
    vg = gs.trace_visibility(seg, N, use_transmittance=True)
    vo = tree.trace_visibility(seg, N, use_transmittance=True)
    assert (np.abs(vg - vo) <= 1e-5).mean() >= 0.999
