"""GPU parity of the extend kernel (closest hit) against the CPU oracle, through the C ABI.
Config C1 of BASELINE.json: primary-ray closest hit on one generated mesh, 512x512, 1 spp."""
import numpy as np
import pytest

from raytracer_amd import _ffi, synth
from tests.helpers import random_rays, compare_hits
from tests.oracle_ffi import OracleScene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c1(rtx):
    sd, cam = synth.single_mesh_scene("teapots", 0.25)
    return sd, cam, rtx.Scene(sd), OracleScene(sd)


@pytest.mark.parametrize("side", [_ffi.HIT_FRONT, _ffi.HIT_BACK, _ffi.HIT_FRONT_AND_BACK])
def test_c1_primary_matches_oracle(c1, side):
    sd, cam, gs, osc = c1
    gp, gt, gf = gs.render_primary(cam, side)
    op, ot, of = osc.render_primary(cam, side, use_bvh=True)
    agree = (gp == op)
    assert agree.mean() >= 0.9999, f"triangle ids agree on {agree.mean():.6f}"
    hit = agree & (gp >= 0)
    assert hit.sum() > 1000
    rel = np.abs(gt[hit] - ot[hit]) / ot[hit]
    assert rel.max() <= 1e-5
    assert (gf[hit] == of[hit]).all()
    # the arithmetic order is the reference's, so t is expected to be bit-identical where ids agree
    assert (gt[hit] == ot[hit]).mean() > 0.999


def test_c1_matches_bruteforce(c1):
    """The reference's own A/B oracle (--nobvh, objects.cpp:256-265): brute force over all faces."""
    sd, cam, gs, osc = c1
    small = _ffi.Camera.from_buffer_copy(cam)
    small.width = small.height = 96
    gp, gt, _ = gs.render_primary(small, _ffi.HIT_FRONT_AND_BACK)
    op, ot, _ = osc.render_primary(small, _ffi.HIT_FRONT_AND_BACK, use_bvh=False)
    assert (gp == op).all()
    assert (gt == ot).all()


def test_random_rays_full_hitinfo(c1):
    sd, cam, gs, osc = c1
    rng = np.random.default_rng(7)
    rays = random_rays(rng, 20000, osc.bounds())
    g = gs.trace_closest(rays, _ffi.HIT_FRONT_AND_BACK)
    o = osc.trace_closest(rays, _ffi.HIT_FRONT_AND_BACK)
    same, hit = compare_hits(g, o, gs.face_base)
    assert hit.sum() > 500
    for k, tol in (("p", 1e-5), ("N", 1e-5), ("GN", 1e-5), ("uvw", 1e-5), ("dpdu", 1e-4), ("dpdv", 1e-4)):
        d = np.abs(g[k][hit] - o[k][hit]).max()
        assert d <= tol * max(1.0, np.abs(o[k][hit]).max()), (k, d)
    assert (g["front"][hit] == o["front"][hit]).all()
    assert (g["mtl_id"][hit] == o["mtl_id"][hit]).all()
    # misses leave t at the caller's bound
    miss = same & ~hit
    assert (g["t"][miss] == rays["tmax"][miss]).all()


def test_tmax_bound_and_empty(c1, rtx):
    sd, cam, gs, osc = c1
    rng = np.random.default_rng(11)
    rays = random_rays(rng, 5000, osc.bounds())
    full = osc.trace_closest(rays, _ffi.HIT_FRONT_AND_BACK)
    hit = full["obj"] >= 0
    rays["tmax"][hit] = full["t"][hit] * np.float32(0.999)   # in-out z upper bound below the hit: must miss
    g = gs.trace_closest(rays, _ffi.HIT_FRONT_AND_BACK)
    o = osc.trace_closest(rays, _ffi.HIT_FRONT_AND_BACK)
    compare_hits(g, o, gs.face_base)
    assert gs.trace_closest(rays[:0]).shape[0] == 0
    # empty scene: every ray misses
    empty = rtx.scene.SceneData()
    es = rtx.Scene(empty)
    e = es.trace_closest(rays[:64], _ffi.HIT_FRONT_AND_BACK)
    assert (e["obj"] == -1).all()


def test_transformed_instances(rtx):
    """Non-identity Node transforms (ToNodeCoords/FromNodeCoords, IntersectNodeAdapter.h:62,93)."""
    from raytracer_amd.scene import SceneData, Mesh
    V, N, UV, F = synth.torus(1.0, 0.3, 24, 12)
    sd = SceneData()
    sd.materials = synth._materials()[:2]
    a = 0.7
    xf = np.array([[1.5 * np.cos(a), -np.sin(a), 0, 0.5], [1.5 * np.sin(a), np.cos(a), 0, -0.25], [0, 0, 0.8, 1.0]], np.float32)
    sd.meshes = [Mesh(V, F, vn=N, vt=UV, material=0), Mesh(V, F, vn=N, vt=UV, material=1, xform=xf.reshape(12))]
    gs, osc = rtx.Scene(sd), OracleScene(sd)
    rays = random_rays(np.random.default_rng(3), 20000, osc.bounds())
    g = gs.trace_closest(rays, _ffi.HIT_FRONT_AND_BACK)
    o = osc.trace_closest(rays, _ffi.HIT_FRONT_AND_BACK)
    # the ray is carried to node coordinates and meets the object's own vertices, as in the reference: the north_star
    # bar (ids >= 99.99 %, t within 1e-5) holds under transforms too; measured: every id equal, t bit-identical
    same, hit = compare_hits(g, o, gs.face_base)
    assert same.all() and np.array_equal(g["t"][hit], o["t"][hit])
    assert np.array_equal(g["bu"][hit], o["bu"][hit]) and np.array_equal(g["bv"][hit], o["bv"][hit])
    for f in ("N", "GN", "p", "uvw", "dpdu", "dpdv"):
        assert np.abs(g[f][hit] - o[f][hit]).max() <= 1e-5 * max(1.0, np.abs(o[f][hit]).max()), f
    assert (g["front"][hit] == o["front"][hit]).all()


def test_degenerate_directions(c1):
    """Axis-parallel rays (1/dir = +-inf) and zero-length directions, as the shade kernel can emit after a degenerate
    reflection: same hits as the oracle, and the box test must still cull (a slab test that turns inf into NaN makes
    such rays walk the whole tree)."""
    sd, cam, gs, osc = c1
    rng = np.random.default_rng(5)
    n = 4096
    rays = random_rays(rng, n, osc.bounds())
    axes = rng.integers(0, 3, n)
    signs = rng.choice(np.array([-1.0, 1.0], np.float32), n)
    d = np.zeros((n, 3), np.float32)
    d[np.arange(n), axes] = signs
    d[::2, (axes[::2] + 1) % 3] = np.float32(0.6)          # half of them parallel to one axis only
    d[1::16] = 0.0                                           # some with no direction at all
    d[np.arange(n) % 7 == 0] = np.where(d[np.arange(n) % 7 == 0] == 0, np.float32(-0.0), d[np.arange(n) % 7 == 0])
    rays["d"] = d / np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-30).astype(np.float32)
    gs.set_counters(True)
    g = gs.trace_closest(rays, _ffi.HIT_FRONT_AND_BACK)
    cnt = gs.get_counters()
    gs.set_counters(False)
    o = osc.trace_closest(rays, _ffi.HIT_FRONT_AND_BACK)
    compare_hits(g, o, gs.face_base)
    nodes = gs.info()["nodes"]
    assert cnt["rays"] == n
    assert cnt["nodes"] / n < 0.02 * nodes + 200, f"{cnt['nodes'] / n:.0f} nodes per ray of {nodes}"
