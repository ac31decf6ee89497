"""Differential fuzzing (tools/fuzz_parity.py) as regression tests: random scenes through the C ABI and through the
oracle. The seeds kept here are the ones that found defects (mirrored transforms baked with the wrong facing, the
tangent frame of a transformed mesh without usable texture coordinates) plus a spread of ordinary ones; run the tool
itself for more."""
import numpy as np
import pytest

from raytracer_amd import _ffi
from raytracer_amd.scene import Light, Material, Mesh, SceneData, TexMap, make_camera, render_params
from tests.oracle_ffi import OracleScene
from tests.test_stress_gpu import check
from tools.fuzz_parity import agree, box_mesh, rand_scene

pytestmark = pytest.mark.gpu

MIRROR = np.array([-0.7, 0.1, 0.0, 0.2, 0.1, 0.8, 0.2, 0.9, 0.0, -0.3, 0.6, -0.4], np.float32)   # det < 0


def grid_rays(n=24):
    ray_dt, _, _ = _ffi.np_dtypes()
    rng = np.random.default_rng(2)
    r = np.zeros(n * n, dtype=ray_dt)
    r["o"] = np.array([0.5, 1.4, 4.0], np.float32) + rng.uniform(-0.3, 0.3, (n * n, 3)).astype(np.float32)
    tgt = np.stack(np.meshgrid(np.linspace(-1.2, 1.6, n), np.linspace(-0.2, 2.0, n)), -1).reshape(-1, 2)
    d = np.concatenate([tgt, np.full((n * n, 1), -0.4)], 1).astype(np.float32) - r["o"]
    r["d"] = d / np.linalg.norm(d, axis=1, keepdims=True)
    r["tmax"] = 1e30
    return r


@pytest.mark.parametrize("with_normals", [False, True])
def test_mirrored_transform_closest_hit(rtx, with_normals):
    """The reference intersects in object space and carries the normal to the world by the inverse transpose
    (IntersectNodeAdapter.h:62,93): under a mirroring transform front/back, GN and the barycentrics must still be the
    object's own, not those of the world-space winding."""
    rng = np.random.default_rng(3)
    sd = SceneData()
    sd.materials = [Material()]
    m = box_mesh(rng, 0, MIRROR)
    if with_normals:
        m.vn = (m.v / np.linalg.norm(m.v, axis=1, keepdims=True) + rng.normal(scale=0.1, size=m.v.shape)).astype(np.float32)
    sd.meshes = [m]
    gs, osc = rtx.Scene(sd), OracleScene(sd)
    rays = grid_rays()
    for side in (_ffi.HIT_FRONT, _ffi.HIT_BACK, _ffi.HIT_FRONT_AND_BACK):
        g, o = gs.trace_closest(rays, hit_side=side), osc.trace_closest(rays, hit_side=side)
        hit = o["obj"] >= 0
        assert hit.sum() > 50 or side == _ffi.HIT_BACK
        same = (g["obj"] == o["obj"]) & (g["face"] == o["face"])
        assert same.mean() >= 0.9999                                   # object-space test on both sides: north_star bar
        k = same & hit
        assert (g["front"][k] == o["front"][k]).all()
        assert np.allclose(g["t"][k], o["t"][k], rtol=1e-5)
        for f in ("GN", "N"):
            assert np.abs(g[f][k] - o[f][k]).max() <= 2e-5, f
        assert np.abs(g["bu"][k] - o["bu"][k]).max() <= 2e-4 and np.abs(g["bv"][k] - o["bv"][k]).max() <= 2e-4
        assert np.abs(g["uvw"][k] - o["uvw"][k]).max() <= 2e-4
        assert np.abs(g["dpdu"][k] - o["dpdu"][k]).max() <= 1e-3 and np.abs(g["dpdv"][k] - o["dpdv"][k]).max() <= 1e-3
    if not with_normals:                                              # seen from outside a closed box every hit is a front hit
        g = gs.trace_closest(rays, hit_side=_ffi.HIT_FRONT_AND_BACK)
        assert (g["front"][g["obj"] >= 0] == 1).all()


def test_tangent_frame_of_transformed_mesh(rtx):
    """Faces whose texture coordinates are degenerate take their tangent frame from the geometric normal — in object
    space, then through the node transform (objects.cpp:233-240). Seen through a bump map."""
    rng = np.random.default_rng(5)
    sd = SceneData()
    bump = rng.uniform(0, 1, (16, 16, 3)).astype(np.float32)
    sd.texmaps = [TexMap(bump)]
    sd.materials = [Material(diffuse=(0.8, 0.7, 0.6), specular=(0.3, 0.3, 0.3), glossiness=50.0, bump_tex=0, bump_scale=1.5)]
    xf = np.array([0.5, 0.2, 0.0, 0.0, -0.1, 0.4, 0.3, 0.8, 0.1, -0.2, 0.6, 0.0], np.float32)
    sd.meshes = [box_mesh(rng, 0, xf), box_mesh(rng, 0, MIRROR + np.array([0, 0, 0, 1.6] + [0] * 8, np.float32))]
    nouv = box_mesh(rng, 0, xf + np.array([0, 0, 0, -1.6] + [0] * 8, np.float32))
    nouv.vt = None                                                    # no texture coordinates at all
    sd.meshes.append(nouv)
    sd.lights = [Light((2.0, 4.0, 3.0), (30, 30, 30), size=0.0)]
    sd.environment = (0.2, 0.2, 0.2)
    gs, osc = rtx.Scene(sd), OracleScene(sd)
    rays = grid_rays()
    g, o = gs.trace_closest(rays, hit_side=_ffi.HIT_FRONT), osc.trace_closest(rays, hit_side=_ffi.HIT_FRONT)
    k = (g["obj"] == o["obj"]) & (g["face"] == o["face"]) & (o["obj"] >= 0)
    assert k.sum() > 100 and len(set(o["obj"][k].tolist())) == 3
    assert np.abs(g["dpdu"][k] - o["dpdu"][k]).max() <= 1e-3 and np.abs(g["dpdv"][k] - o["dpdv"][k]).max() <= 1e-3
    cam = make_camera((0.5, 1.6, 4.5), (0.2, 0.8, 0), fov=45, width=48, height=32)
    for mode in (_ffi.SHADING_PATH, _ffi.SHADING_LEGACY):
        rp = render_params(spp_min=4, max_bounces=3, shading_mode=mode)
        gi, _ = gs.render_samples(cam, rp)
        oi = osc.render_samples(cam, rp)
        check(gi, oi, mode)


@pytest.mark.parametrize("seed", [5, 8, 12, 28, 32, 33, 36, 37, 39, 61, 77, 101, 150, 199])
def test_random_scene(rtx, seed):
    sd, cam, rng = rand_scene(seed)
    gs, osc = rtx.Scene(sd), OracleScene(sd)
    for mode in (_ffi.SHADING_PATH, _ffi.SHADING_LEGACY):
        rp = render_params(spp_min=4, max_bounces=int(rng.choice([1, 3, 8])), seed=seed * 7919 + 1, shading_mode=mode)
        g, _ = gs.render_samples(cam, rp)
        o = osc.render_samples(cam, rp)
        assert np.isfinite(g).all() or not np.isfinite(o).all()
        check(g, o, mode)
        assert np.array_equal(g[..., 3], o[..., 3])


def test_constant_bright_pixels_are_not_refined(rtx):
    """A lamp that fills the view: every sample is the clamped radiance 10, the variance is exactly zero and the
    confidence interval test must stop at spp_min (RayCaster.cpp:848-868). A quotient that is a few ulp off makes
    s2/n - (s1/n)^2 positive and the half width 0.01 > 0.005."""
    sd = SceneData()
    sd.materials = [Material()]
    sd.meshes = [Mesh(np.array([[-9, -9, -5], [9, -9, -5], [0, 9, -5]], np.float32), [[0, 1, 2]], material=0)]
    sd.lights = [Light((0.0, 0.0, 0.0), (3000.0, 3000.0, 3000.0), size=1.0)]
    sd.environment = (0.3, 0.3, 0.3)
    cam = make_camera((0, 0, 4), (0, 0, 0), fov=20, width=24, height=16)
    gs, osc = rtx.Scene(sd), OracleScene(sd)
    for spp_min in (3, 5, 7):
        rp = render_params(spp_min=spp_min, spp_max=4 * spp_min, ci_halfwidth=0.005, max_bounces=4)
        g, o = gs.render(cam, rp), osc.render(cam, rp)
        assert (o["samples"][4:12, 8:16] == spp_min).all()           # the oracle stops there
        assert np.array_equal(g["samples"], o["samples"])
        assert np.abs(g["linear"] - o["linear"]).max() <= 2e-5 * 10


def test_duplicate_faces_tie_rule(rtx):
    """Exact duplicates inside a mesh and a second copy of the whole object: the smaller global face id wins on the
    GPU, in the oracle's BVH walk and in its brute-force loop alike."""
    rng = np.random.default_rng(13)
    a = box_mesh(rng, 0, np.array([0.6, 0, 0, 0.1, 0, 0.5, 0, 0.7, 0, 0, 0.7, 0.0], np.float32))
    a.f = np.ascontiguousarray(np.concatenate([a.f, a.f[:4], a.f[2:3]]))
    a.face_mtl = np.arange(a.f.shape[0], dtype=np.int32)
    b = box_mesh(rng, 1, a.xform.copy())
    sd = SceneData()
    sd.materials = [Material(), Material(diffuse=(0.1, 0.9, 0.1))]
    sd.meshes = [a, b]
    gs, osc = rtx.Scene(sd), OracleScene(sd)
    rays = grid_rays(32)
    for side in (_ffi.HIT_FRONT, _ffi.HIT_BACK, _ffi.HIT_FRONT_AND_BACK):
        g = gs.trace_closest(rays, hit_side=side)
        o = osc.trace_closest(rays, hit_side=side, use_bvh=True)
        ob = osc.trace_closest(rays, hit_side=side, use_bvh=False)
        assert (o["face"] == ob["face"]).all() and (o["obj"] == ob["obj"]).all()
        same = (g["obj"] == o["obj"]) & (g["face"] == o["face"]) & (g["mtl_id"] == o["mtl_id"])
        assert same.mean() >= 0.9999
        hit = g["obj"] >= 0
        assert (g["obj"][hit] == 0).all() and (g["face"][hit] < 12).all()


@pytest.mark.parametrize("seed,spice,big,spp", [(20, True, False, 4), (44, False, True, 4), (63, False, True, 4), (332, False, True, 24)])
def test_rounding_cases_found_at_the_bit_identity_bar(rtx, seed, spice, big, spp):
    """One sample in 10^5 .. 10^6 of these scenes used to part from the oracle (tools/fuzz_parity.py --exact):
    seeds 20 / 44: a shadow hop that ended early at an opaque blocker although a surface that lets light through lay in
    front of it (the next hop starts 1e-4 off the ray and slips past the blocker's edge) - the early exit now looks for
    such surfaces before it answers; seed 63: a camera ray on the very edge of a Plane, which the reference culls by its
    box test (scene_box.cpp:8-46) before Plane::IntersectRay sees it; seed 332: the same for a shadow ray on the edge of
    a box-shaped mesh (IntersectNodeAdapter.h:63-75). Path mode: every sample bit-identical."""
    from tools.fuzz_parity import spice_scene
    from tests.helpers import assert_path_samples_identical
    sd, cam, rng = rand_scene(seed)
    if spice:
        spice_scene(sd, cam, seed)
    if big:
        cam.width, cam.height = 160 + 8 * (seed % 7), 120 + 4 * (seed % 5)
    gs, osc = rtx.Scene(sd), OracleScene(sd)
    rp = render_params(spp_min=spp, max_bounces=3, seed=seed * 7919 + 1)
    g, _ = gs.render_samples(cam, rp)
    assert_path_samples_identical(g, osc.render_samples(cam, rp))
