"""The C ABI under misuse: unknown enum values, out-of-range indices, non-finite parameters and buffers that are too
small are rejected with an error code and a message (nothing is launched); degenerate but legal calls (no rays, no
pixels, no photons) succeed and leave their outputs alone."""
import ctypes as C

import numpy as np
import pytest

from raytracer_amd import _ffi
from raytracer_amd.scene import Light, Material, Mesh, SceneData, TexMap, make_camera, render_params
from tests.test_stress_gpu import cam_for, room

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scene(rtx):
    sd = room()
    return sd, rtx.Scene(sd), cam_for(1.0, 40, 24)


def test_render_params_rejected(scene, rtx):
    sd, gs, cam = scene
    bad = [
        (dict(shading_mode=7), "shading_mode"),
        (dict(filter_type=99), "filter_type"),
        (dict(filter_type=-1), "filter_type"),
        (dict(filter_radius=1e9), "filter_radius"),
        (dict(filter_radius=float("inf")), "filter_radius"),
        (dict(filter_radius=float("nan")), "filter_radius"),
        (dict(max_bounces=0), "max_bounces"),
        (dict(spp_max=1), "spp"),                         # below spp_min
        (dict(index_stride=1), "index_stride"),
        (dict(rank=-1), "rank"),
    ]
    for kw, word in bad:
        args = dict(spp_min=2, max_bounces=4)
        args.update(kw)
        with pytest.raises(rtx.RtxError, match=word):
            gs.render(cam, render_params(**args))
    # and the scene still renders afterwards
    img = gs.render(cam, render_params(spp_min=2, max_bounces=4))
    assert np.isfinite(img["linear"]).all() and img["linear"].max() > 0


def test_bad_camera_rejected(scene, rtx):
    sd, gs, cam = scene
    rp = render_params(spp_min=1, max_bounces=2)
    flat = make_camera((0, 1, 4), (0, 1, 0), width=0, height=10)
    with pytest.raises(rtx.RtxError, match="camera"):
        gs.render_samples(flat, rp)
    par = make_camera((0, 1, 4), (0, 1, 0), width=8, height=8)
    par.up = (C.c_float * 3)(*list(par.dir))
    with pytest.raises(rtx.RtxError, match="parallel"):
        gs.render_samples(par, rp)


def test_trace_entry_points_rejected(scene, rtx):
    sd, gs, cam = scene
    ray_dt, _, _ = _ffi.np_dtypes()
    rays = np.zeros(4, dtype=ray_dt)
    rays["d"] = (0, 0, -1)
    rays["tmax"] = 1e30
    with pytest.raises(rtx.RtxError, match="hit_side"):
        gs.trace_closest(rays, hit_side=7)
    with pytest.raises(rtx.RtxError, match="hit_side"):
        gs.render_primary(cam, hit_side=-2)
    assert len(gs.trace_closest(rays[:0])) == 0                      # no rays: fine, nothing launched
    at = _ffi.ShadePoint()
    at.P = (C.c_float * 3)(0, 0.5, 0)
    at.N = (C.c_float * 3)(0, 1, 0)
    dirs = np.array([[0, 1, 0]], np.float32)
    for child in (0, 4):
        with pytest.raises(rtx.RtxError, match="child"):
            gs.trace_secondary(at, dirs, child=child)
    at.diffuse_depth = 300
    with pytest.raises(rtx.RtxError, match="depth"):
        gs.trace_secondary(at, dirs)
    at.diffuse_depth = 0
    assert gs.trace_secondary(at, dirs[:0]).shape[0] == 0
    out = gs.trace_secondary(at, np.array([[0, 0, 0], [np.nan, 0, 1], [0, 1, 0]], np.float32))
    assert np.isfinite(out[2]).all()                                 # a zero / NaN direction does not disturb its neighbours


def test_render_round_lists(scene, rtx):
    """Out-of-range pixel indices are refused before anything is launched; an empty list is a no-op; a duplicate-free
    ragged list accumulates only its pixels."""
    import torch
    sd, gs, cam = scene
    W, H = cam.width, cam.height
    dev = torch.device("cuda:0")
    accum = torch.zeros(W * H * 4, dtype=torch.float32, device=dev)
    count = torch.zeros(W * H, dtype=torch.int32, device=dev)
    z = torch.full((W * H,), 3.0e38, dtype=torch.float32, device=dev)
    s1 = torch.zeros(W * H * 4, dtype=torch.float32, device=dev)
    s2 = torch.zeros(W * H * 4, dtype=torch.float32, device=dev)
    film = _ffi.Film()
    film.accum, film.count, film.z = accum.data_ptr(), count.data_ptr(), z.data_ptr()
    ps = _ffi.PixelStats()
    ps.sum, ps.sum2 = s1.data_ptr(), s2.data_ptr()
    rp = render_params(spp_min=2, spp_max=8, max_bounces=3)
    for badlist in ([W * H], [-1], [0, 5, W * H + 7]):
        with pytest.raises(rtx.RtxError, match="outside the frame"):
            gs.render_round(cam, rp, film, ps, badlist, 0, 2)
    torch.cuda.synchronize()
    assert int(count.sum()) == 0 and float(accum.abs().sum()) == 0.0
    with pytest.raises(rtx.RtxError, match="n_samples"):
        gs.render_round(cam, rp, film, ps, [0], 0, 0)
    with pytest.raises(rtx.RtxError, match="sample_base"):
        gs.render_round(cam, rp, film, ps, [0], -1, 2)
    st = gs.render_round(cam, rp, film, ps, [], 0, 2)
    torch.cuda.synchronize()
    assert st["rays_primary"] == 0 and int(count.sum()) == 0
    picks = np.array([0, 1, W + 3, W * H - 1], np.int32)
    st = gs.render_round(cam, rp, film, ps, picks, 0, 2)
    torch.cuda.synchronize()
    assert st["rays_primary"] == 2 * len(picks)
    c = count.cpu().numpy()
    assert c.sum() == 2 * len(picks) and (c[picks] == 2).all()
    # refine list into a buffer that is too small: the count is still the full one, only `capacity` entries are written
    lib = gs.lib
    from raytracer_amd.api import refine_list
    full = refine_list(ps, W, H, -1.0, 8)           # negative halfwidth: every pixel sampled 2 .. spp_max-1 times refines
    assert full.tolist() == sorted(picks.tolist())
    small = np.full(3, -7, np.int32)
    n = lib.rtx_refine_list(C.byref(ps), W, H, C.c_float(-1.0), 8, small.ctypes.data_as(_ffi.c_int_p), 2)
    assert n == len(full) and (small[:2] == full[:2]).all() and small[2] == -7
    assert lib.rtx_refine_list(C.byref(ps), W, H, C.c_float(-1.0), 8, None, 0) == len(full)
    assert lib.rtx_refine_list(C.byref(ps), 0, H, C.c_float(-1.0), 8, None, 0) < 0


def _desc_error(rtx, sd, word):
    with pytest.raises(rtx.RtxError, match=word):
        rtx.Scene(sd)


def test_scene_descriptions_rejected(rtx):
    tri = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    def base():
        sd = SceneData()
        sd.materials = [Material()]
        sd.meshes = [Mesh(tri.copy(), [[0, 1, 2]], material=0)]
        sd.lights = [Light((0, 3, 0), (5, 5, 5))]
        return sd
    sd = base(); sd.meshes[0].f[0, 2] = 3
    _desc_error(rtx, sd, "vertex index")
    sd = base(); sd.meshes[0].f[0, 0] = -1
    _desc_error(rtx, sd, "vertex index")
    sd = base(); sd.meshes[0].vn = np.array([[0, 0, 1]], np.float32)       # vn indexed through f: 1 and 2 are out of range
    _desc_error(rtx, sd, "normal index")
    sd = base(); sd.meshes[0].vt = np.array([[0, 0, 0]], np.float32); sd.meshes[0].ft = np.array([[0, 0, 5]], np.int32)
    _desc_error(rtx, sd, "texcoord index")
    sd = base(); sd.meshes[0].material = 1
    _desc_error(rtx, sd, "material index")
    sd = base(); sd.meshes[0].kind = 9
    _desc_error(rtx, sd, "kind")
    sd = base(); sd.meshes[0].v[1, 0] = np.nan
    _desc_error(rtx, sd, "non-finite vertex")
    sd = base(); sd.meshes[0].xform[3] = np.inf
    _desc_error(rtx, sd, "non-finite transform")
    sd = base(); sd.meshes[0].v[:] *= 1e30; sd.meshes[0].xform[0] = 1e30
    _desc_error(rtx, sd, "overflows")
    sd = base(); sd.meshes.append(Mesh.sphere(xform=np.array([np.nan] + [0] * 11, np.float32), material=0))
    _desc_error(rtx, sd, "non-finite transform")
    sd = base(); sd.materials[0].type = 5
    _desc_error(rtx, sd, "texture index or type")
    sd = base(); sd.materials[0].diffuse_tex = 0                           # no texture maps in the scene
    _desc_error(rtx, sd, "texture index or type")
    sd = base(); sd.materials[0].bump_tex = -2
    _desc_error(rtx, sd, "texture index or type")
    sd = base(); sd.environment_tex = 3
    _desc_error(rtx, sd, "environment/background")
    sd = base(); sd.lights = [Light((0, 3, 0), (1, 1, 1))] * 9
    _desc_error(rtx, sd, "RTX_MAX_LIGHTS")
    sd = base(); sd.texmaps = [TexMap(np.zeros((0, 4, 3), np.uint8))]
    _desc_error(rtx, sd, "empty")
    sd = base(); sd.texmaps = [TexMap(np.zeros((2, 2, 3), np.uint8))]; sd.texmaps[0].format = 4
    _desc_error(rtx, sd, "texture format")
    # after all of that a good description still builds and traces
    gs = rtx.Scene(base())
    ray_dt, _, _ = _ffi.np_dtypes()
    r = np.zeros(1, dtype=ray_dt)
    r["o"], r["d"], r["tmax"] = (0.2, 0.2, 1), (0, 0, -1), 1e30
    assert abs(gs.trace_closest(r)["t"][0] - 1.0) < 1e-6


def test_photon_calls_rejected(scene, rtx):
    sd, gs, cam = scene
    for kw, word in ((dict(num_photons=-1), "photon parameters"), (dict(max_bounces=4), "photon parameters"),
                     (dict(first_photon=-3), "first_photon"), (dict(photon_radius=float("nan")), "finite"),
                     (dict(caustic_radius=float("inf")), "finite")):
        args = dict(num_photons=1000)
        args.update(kw)
        with pytest.raises(rtx.RtxError, match=word):
            rtx.Photons(gs, **args)
    gp = rtx.Photons(gs, num_photons=2000)
    lib = gs.lib
    assert lib.rtx_photon_download(gp.h, 2, None, 0) < 0
    assert lib.rtx_photon_download(gp.h, 0, None, 0) == 0                  # capacity 0: nothing to write
    if gp.info()["n_global"]:
        assert lib.rtx_photon_download(gp.h, 0, None, 10) < 0              # something to write and nowhere to put it
    assert lib.rtx_photon_estimate(gp.h, 0, None, None, 5, None) < 0
    assert lib.rtx_photon_estimate(gp.h, 0, None, None, 0, None) == 0
    assert lib.rtx_photon_set(gp.h, 3, None, 0, 1) < 0
    assert lib.rtx_photon_set(gp.h, 0, None, 4, 1) < 0


def test_photons_outside_the_scene_box(scene, rtx):
    """Photons handed in through rtx_photon_set may lie outside the scene box the hash grid spans (they are clamped
    into its border cells): an estimate next to them must still find them. Checked against the closed form of the
    cone-filtered estimate with fewer than K photons in range: 3 / (pi r^2) * sum(power * (1 - d / r))."""
    import torch
    sd, gs, cam = scene
    gp = rtx.Photons(gs, num_photons=0, caustic_radius=0.25)
    r = gp.info()["caustic_radius"]
    assert abs(r - 0.25) < 1e-7
    centre = np.array([40.0, -30.0, 55.0], np.float32)                     # the room spans +-4
    rng = np.random.default_rng(5)
    n = 20
    pos = centre + rng.uniform(-0.1, 0.1, (n, 3)).astype(np.float32)
    rec = np.zeros((n, 12), np.float32)
    rec[:, 0:3] = pos
    rec[:, 3] = np.arange(n, dtype=np.uint32).view(np.float32)             # photon index (tie-break key)
    rec[:, 4:7] = (0, -1, 0)
    rec[:, 8:11] = rng.uniform(0.5, 2.0, (n, 3)).astype(np.float32)
    buf = torch.from_numpy(rec).to("cuda:0")
    # total_emitted = 10 n: the caustics map holds total_emitted / 10 records (PhotonMapBuilder.cpp:28)
    assert gs.lib.rtx_photon_set(gp.h, 1, buf.data_ptr(), n, 10 * n) == 0
    q = np.stack([centre, centre + np.float32(0.3), np.array([0, 0.5, 0], np.float32)])
    nrm = np.tile(np.array([[0, 1, 0]], np.float32), (len(q), 1))
    got = gp.estimate(1, q, nrm)
    want = np.zeros_like(got)
    for i, p in enumerate(q):
        d = np.sqrt(((pos - p) ** 2).sum(-1))
        m = d <= r
        want[i] = (rec[m, 8:11] / n * (1 - d[m, None] / r)).sum(0) * 3 / (np.pi * r * r)
    assert want[0].min() > 0 and np.allclose(got, want, rtol=2e-4, atol=1e-7)
    assert (got[2] == 0).all()                                             # inside the room there are none
    assert (gp.estimate(1, q, -nrm) == 0).all()                            # photons arriving from behind do not count
