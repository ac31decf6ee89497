"""Odd shapes and extreme settings, GPU against the oracle through the C ABI: image sizes that are not multiples of the
bucket or of 64 pixels, 1x1 images, no lights, eight lights (more shadow jobs than the tail kernel's quad holds in one
round), an all-glass scene, tiny and huge coordinates, one-bounce limits, both shading modes."""
import numpy as np
import pytest

from raytracer_amd import _ffi
from raytracer_amd.scene import Light, Material, Mesh, SceneData, make_camera, render_params
from tests.helpers import assert_path_samples_identical, assert_legacy_samples_close
from tests.oracle_ffi import OracleScene

pytestmark = pytest.mark.gpu


def box(lo, hi, material):
    lo, hi = np.asarray(lo, np.float32), np.asarray(hi, np.float32)
    c = np.array([[lo[0] if not i & 1 else hi[0], lo[1] if not i & 2 else hi[1], lo[2] if not i & 4 else hi[2]] for i in range(8)], np.float32)
    f = [[0, 2, 1], [1, 2, 3], [4, 5, 6], [5, 7, 6], [0, 1, 4], [1, 5, 4], [2, 6, 3], [3, 6, 7], [0, 4, 2], [2, 4, 6], [1, 3, 5], [3, 7, 5]]
    return Mesh(c, f, material=material)


def room(n_lights=3, scale=1.0, glass_only=False):
    sd = SceneData()
    sd.materials = [
        Material(diffuse=(0.7, 0.7, 0.65), specular=(0.2, 0.2, 0.2), glossiness=40.0),
        Material(diffuse=(0.0, 0.0, 0.0), specular=(0.9, 0.9, 0.9), refraction=(0.95, 0.95, 0.95), ior=1.5, glossiness=1e4,
                 absorption=(0.3, 0.05, 0.05)),
        Material(diffuse=(0.1, 0.1, 0.1), specular=(0.6, 0.6, 0.6), reflection=(0.8, 0.8, 0.8), glossiness=500.0),
    ]
    s = np.float32(scale)
    floor = np.array([[-4, 0, -4], [4, 0, -4], [4, 0, 4], [-4, 0, 4]], np.float32) * s
    sd.meshes = [Mesh(floor, [[0, 2, 1], [0, 3, 2]], material=1 if glass_only else 0)]
    sd.meshes.append(box(np.array([-1.5, 0.0, -0.5]) * s, np.array([-0.5, 1.0, 0.5]) * s, 1))
    sd.meshes.append(box(np.array([0.4, 0.0, -0.6]) * s, np.array([1.4, 0.8, 0.4]) * s, 1 if glass_only else 2))
    sd.meshes.append(Mesh.sphere(xform=np.array([0.4 * s, 0, 0, 0.0, 0, 0.4 * s, 0, 1.6 * s, 0, 0, 0.4 * s, 0.3 * s], np.float32),
                                 material=1 if glass_only else 0))
    rng = np.random.default_rng(4)
    sd.lights = [Light(tuple(np.array([rng.uniform(-3, 3), rng.uniform(2.5, 4), rng.uniform(-3, 3)]) * s),
                       tuple(rng.uniform(4, 12, 3) * s * s), size=(0.3 * s if k % 2 == 0 else 0.0)) for k in range(n_lights)]
    sd.environment = (0.15, 0.18, 0.22)
    return sd


def cam_for(scale, w, h):
    return make_camera((0.3 * scale, 1.7 * scale, 4.5 * scale), (0, 0.5 * scale, 0), fov=40, focaldist=4.5 * scale, dof=0.0, width=w, height=h)


def agree(g, o, tol=1e-3):
    d = np.abs(g[..., :3] - o[..., :3]).max(-1)
    return float((d <= tol * (1 + np.abs(o[..., :3]).max(-1))).mean())


def check(g, o, mode=_ffi.SHADING_PATH):
    """Path mode: bit-identical to the oracle. Legacy mode: the kernel sums a sample's recursion tree in another order."""
    if mode == _ffi.SHADING_PATH:
        assert_path_samples_identical(g, o)
    else:
        assert_legacy_samples_close(g, o)


@pytest.mark.parametrize("w,h", [(1, 1), (7, 5), (33, 17), (70, 50)])
def test_odd_image_sizes(rtx, w, h):
    sd = room()
    gs, osc = rtx.Scene(sd), OracleScene(sd)
    cam = cam_for(1.0, w, h)
    rp = render_params(spp_min=3, max_bounces=16)
    g, st = gs.render_samples(cam, rp)
    o = osc.render_samples(cam, rp)
    assert g.shape == o.shape == (h, w, 3, 4) and st["rays_primary"] == w * h * 3
    assert np.isfinite(g).all()
    check(g, o)
    img, ref = gs.render(cam, rp), osc.render(cam, rp)
    assert np.array_equal(img["samples"], ref["samples"])
    assert np.allclose(img["linear"], ref["linear"], rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("mode", [_ffi.SHADING_PATH, _ffi.SHADING_LEGACY])
@pytest.mark.parametrize("n_lights", [0, 1, 8])
def test_light_counts(rtx, mode, n_lights):
    """0 lights: no shadow queue at all; 8 lights: three rounds of shadow jobs per vertex in the tail kernel."""
    sd = room(n_lights)
    gs, osc = rtx.Scene(sd), OracleScene(sd)
    cam = cam_for(1.0, 48, 32)
    rp = render_params(spp_min=4, max_bounces=12, shading_mode=mode)
    g, st = gs.render_samples(cam, rp)
    o = osc.render_samples(cam, rp)
    assert np.isfinite(g).all()
    check(g, o, mode)
    assert (st["rays_shadow"] == 0) == (n_lights == 0)


@pytest.mark.parametrize("scale", [1e-3, 1.0, 1e2, 3e3])
def test_scene_scale(rtx, scale):
    """Millimetre-sized to kilometre-sized copies of the same room: the quantised node frames, the 1e-4 offsets and the
    biases are absolute lengths in the reference, so the images differ between scales - GPU and oracle must not.
    At 3e3 the coordinates reach 1.2e4, where one fp32 ulp (1e-3) is larger than the reference's 1e-4 ray offsets: rays
    start on their own surface and every second hit is decided by the last bit of t (shadow acne, in the reference
    too). There only the well-conditioned part is compared: primary hits, finiteness, overall level."""
    sd = room(3, scale)
    gs, osc = rtx.Scene(sd), OracleScene(sd)
    cam = cam_for(scale, 40, 28)
    for mode in (_ffi.SHADING_PATH, _ffi.SHADING_LEGACY):
        rp = render_params(spp_min=3, max_bounces=12, shading_mode=mode)
        g, _ = gs.render_samples(cam, rp)
        o = osc.render_samples(cam, rp)
        assert np.isfinite(g).all()
        assert np.array_equal(g[..., 3], o[..., 3])
        check(g, o, mode)   # ill-conditioned or not, both sides make the same last-bit decisions


def test_all_glass_long_paths(rtx):
    """Every surface transmissive: paths run to the bounce limit, shadow chains to their 8-hop limit."""
    sd = room(2, 1.0, glass_only=True)
    gs, osc = rtx.Scene(sd), OracleScene(sd)
    cam = cam_for(1.0, 40, 28)
    for mb in (1, 2, 64):
        rp = render_params(spp_min=4, max_bounces=mb)
        g, st = gs.render_samples(cam, rp)
        o = osc.render_samples(cam, rp)
        assert np.isfinite(g).all()
        check(g, o)
    assert st["rays_bounce"] > st["rays_primary"] // 2 and st["rays_shadow"] > 0


def test_empty_and_degenerate_scenes(rtx):
    """No geometry at all, and geometry that no ray can hit (zero-area triangles): every path leaves for the
    environment at once; nothing fails, GPU and oracle agree exactly."""
    for meshes in ([], [Mesh(np.array([[0, 0, 0], [1, 1, 1], [2, 2, 2], [1, 1, 1]], np.float32), [[0, 1, 2], [1, 1, 1], [0, 3, 1]], material=0)]):
        sd = SceneData()
        sd.materials = [Material()]
        sd.meshes = meshes
        sd.lights = [Light((0, 3, 0), (5, 5, 5), size=0.0)]     # a point light: nothing for a camera ray to hit
        sd.environment = (0.2, 0.3, 0.4)
        gs, osc = rtx.Scene(sd), OracleScene(sd)
        cam = make_camera((0, 1, 5), (0, 0, 0), fov=50, width=20, height=12)
        for mode in (_ffi.SHADING_PATH, _ffi.SHADING_LEGACY):
            rp = render_params(spp_min=2, max_bounces=8, shading_mode=mode)
            g, st = gs.render_samples(cam, rp)
            o = osc.render_samples(cam, rp)
            assert np.array_equal(g, o) and np.allclose(g[..., :3], [0.2, 0.3, 0.4]) and (g[..., 3] > 1e29).all()
            assert st["rays_bounce"] == 0 and st["rays_shadow"] == 0
        rays = np.zeros(3, _ffi.np_dtypes()[0])
        rays["o"], rays["d"], rays["tmax"] = (0, 0, 5), (0, 0, -1), 1e30
        assert (gs.trace_closest(rays, _ffi.HIT_FRONT_AND_BACK)["obj"] == -1).all()


def test_adaptive_rounds_with_ragged_limits(rtx):
    """spp_max that is no multiple of spp_min (the last round is shorter), a wide Gaussian filter on an image smaller
    than its support, one sample per pass."""
    sd = room()
    gs, osc = rtx.Scene(sd), OracleScene(sd)
    cam = cam_for(1.0, 19, 13)
    rp = render_params(spp_min=3, spp_max=11, ci_halfwidth=0.02, max_bounces=6, filter_type=_ffi.FILTER_GAUSSIAN, filter_radius=3.0,
                       samples_per_pass=1)
    g, o = gs.render(cam, rp), osc.render(cam, rp)
    assert set(np.unique(o["samples"])) <= {3, 6, 9, 11} and (o["samples"] == 11).any()
    assert (g["samples"] == o["samples"]).mean() >= 0.98
    same = g["samples"] == o["samples"]
    assert np.abs(g["linear"] - o["linear"])[same].mean() <= 0.02 * o["linear"].mean()


def test_non_finite_rays_are_misses(rtx):
    """NaN / infinite origins, directions and limits: 'no hit' for that ray, the rest of the batch unaffected."""
    sd = room()
    gs, osc = rtx.Scene(sd), OracleScene(sd)
    rays = np.zeros(8, _ffi.np_dtypes()[0])
    rays["o"], rays["d"], rays["tmax"] = (0.1, 3.0, 0.2), (0, -1, 0), 1e30
    rays["o"][1, 0] = np.nan
    rays["d"][2] = (np.nan, -1, 0)
    rays["d"][3] = (0, 0, 0)
    rays["o"][4, 1] = np.inf
    rays["tmax"][5] = 0.0
    rays["tmax"][6] = -1.0
    g = gs.trace_closest(rays, _ffi.HIT_FRONT_AND_BACK)
    o = osc.trace_closest(rays, _ffi.HIT_FRONT_AND_BACK)
    assert np.array_equal(g["obj"], o["obj"]) and g["obj"][0] >= 0 and g["obj"][7] >= 0
    assert (g["obj"][1:7] == -1).all()
    assert np.array_equal(g["t"][[0, 7]], o["t"][[0, 7]])


def test_photons_without_sources_or_receivers(rtx):
    """No light to emit from, and a scene without a single diffuse surface to store photons on: empty maps, and a render
    that uses them changes nothing."""
    for n_lights, glass_only in ((0, False), (2, True)):
        sd = room(n_lights, 1.0, glass_only)
        gs = rtx.Scene(sd)
        ph = rtx.Photons(gs, num_photons=20000)
        info = ph.info()
        assert info["n_global"] == 0 and info["n_caustic"] == 0
        cam = cam_for(1.0, 24, 16)
        a, _ = gs.render_samples(cam, render_params(spp_min=2, max_bounces=6, use_photons=1), ph)
        b, _ = gs.render_samples(cam, render_params(spp_min=2, max_bounces=6))
        assert np.array_equal(a, b)
