"""GPU parity of photon mapping (config C5): emission + tracing (PhotonMapBuilder.cpp:80-268), the hash-grid
k-nearest estimate (RayCaster.cpp:143-153) and the caustic term of the path loop (:508-522), against the CPU oracle."""
import os

import numpy as np
import pytest

from raytracer_amd import synth
from raytracer_amd.scene import render_params
from tests.helpers import assert_path_samples_identical
from tests.oracle_ffi import OracleScene, OraclePhotons

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def env(rtx):
    sd = synth.build_scene(0.003)
    gs, osc = rtx.Scene(sd), OracleScene(sd)
    N = 300000
    return sd, gs, osc, rtx.Photons(gs, num_photons=N), OraclePhotons(osc, N), N


def _sorted(p):
    """Records in a canonical order (the GPU stores them as its lanes finish, the oracle in emission order)."""
    k = np.lexsort((p["power"][:, 0], p["dir"][:, 0], p["pos"][:, 2], p["pos"][:, 1], p["pos"][:, 0]))
    return p[k]


def test_photon_sets_match(env):
    """Both sides evaluate the same elementary functions (rtx_math.h), so every photon takes the same path: the stored
    records are the same set, bit for bit (the GPU's are unscaled, the oracle's scaled by PhotonMapBuilder.cpp:38-50)."""
    sd, gs, osc, gp, op, N = env
    gi, oi = gp.info(), op.info()
    assert gi["photon_radius"] == oi["photon_radius"] and gi["caustic_radius"] == oi["caustic_radius"]
    assert gi["n_global"] == oi["n_global"] and gi["n_caustic"] == oi["n_caustic"]
    assert oi["n_caustic"] > 100 and oi["n_global"] > 100
    assert gp.stats["rays_photon"] == oi["rays"]
    for which in (0, 1):
        a, b = _sorted(gp.download(which)), _sorted(op.download(which))
        scale = np.float32(1.0) / np.float32(N if which == 0 else max(1, len(b)))
        assert np.array_equal(a["pos"], b["pos"]) and np.array_equal(a["dir"], b["dir"])
        assert np.array_equal(a["power"] * scale, b["power"])


def test_irradiance_estimate(env):
    """k nearest within the radius, cone filter: the lattice and the order of the sum are defined by the photon records
    alone (DESIGN.md "Photon gather"), so the estimates are bit-identical."""
    sd, gs, osc, gp, op, N = env
    rng = np.random.default_rng(5)
    for which in (0, 1):
        b = op.download(which)
        probe = (b["pos"][:: max(1, len(b) // 300)][:300] + np.float32(0.003)).astype(np.float32)
        far = (probe[:20] + np.float32(50.0)).astype(np.float32)          # outside the grid window: nothing in reach
        probe = np.concatenate([probe, far, b["pos"][:20].astype(np.float32)])   # ... and exactly on a photon
        nrm = rng.normal(size=probe.shape).astype(np.float32)
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        ge, oe = gp.estimate(which, probe, nrm), op.estimate(which, probe, nrm)
        assert np.isfinite(ge).all() and np.abs(oe).max() > 0
        assert np.array_equal(ge.view(np.uint32), oe.view(np.uint32))
        assert not ge[300:320].any()


def test_caustic_term_in_path_loop(env):
    sd, gs, osc, gp, op, N = env
    cam = synth.cerealbar_camera(64, 36)
    rp = render_params(spp_min=4, max_bounces=4, use_photons=1)
    g, _ = gs.render_samples(cam, rp, gp)
    o = osc.render_samples(cam, rp, op)
    assert_path_samples_identical(g, o)
    base = osc.render_samples(cam, render_params(spp_min=4, max_bounces=4))
    assert np.abs(o[..., :3] - base[..., :3]).max() > 1e-4      # the caustic map does contribute


def test_photon_split_and_set(env, rtx):
    """Multi-GPU protocol on one GPU: two strided halves, concatenated with rtx_photon_set, equal the full build."""
    import ctypes as C
    import torch
    sd, gs, osc, gp, op, N = env
    a = rtx.Photons(gs, num_photons=N, first_photon=0, photon_stride=2)
    b = rtx.Photons(gs, num_photons=N, first_photon=1, photon_stride=2)
    full = gp.info()
    assert a.info()["n_global"] + b.info()["n_global"] == full["n_global"]
    assert a.info()["n_caustic"] + b.info()["n_caustic"] == full["n_caustic"]
    lib = gs.lib
    for which, key in ((0, "n_global"), (1, "n_caustic")):
        na, nb = a.info()[key], b.info()[key]
        buf = torch.empty((na + nb, 12), dtype=torch.float32, device="cuda:0")
        for src, off, n in ((a, 0, na), (b, na, nb)):
            if n:
                ptr = lib.rtx_photon_device_ptr(src.h, which)
                view = (C.c_float * (12 * n)).from_address  # noqa: F841  (device memory: copy with torch below)
                tmp = torch.empty((n, 12), dtype=torch.float32, device="cuda:0")
                # device-to-device copy through hipMemcpy via torch's raw pointer API
                torch.cuda.synchronize()
                import ctypes
                hip = ctypes.CDLL("libamdhip64.so")
                assert hip.hipMemcpy(ctypes.c_void_p(tmp.data_ptr()), ctypes.c_void_p(ptr), ctypes.c_size_t(48 * n), 3) == 0
                buf[off:off + n] = tmp
        torch.cuda.synchronize()
        assert lib.rtx_photon_set(a.h, which, buf.data_ptr(), na + nb, N) == 0
    probe = op.download(1)["pos"][:100].astype(np.float32)
    nrm = np.tile(np.array([[0, 1, 0]], np.float32), (len(probe), 1))
    for which in (0, 1):
        e1, e2 = a.estimate(which, probe, nrm), gp.estimate(which, probe, nrm)
        assert np.array_equal(e1.view(np.uint32), e2.view(np.uint32))   # same records -> same (cell, key) order -> same bits


def test_golden_photon_fixture(rtx):
    g = np.load(os.path.join(GOLD, "photon_small.npz"))
    sd = synth.build_scene(0.002)
    gs = rtx.Scene(sd)
    gp = rtx.Photons(gs, num_photons=100000)
    info = gp.info()
    assert info["n_global"] == int(g["n_global"]) and info["n_caustic"] == int(g["n_caustic"])
    assert np.array_equal(np.float32([info["photon_radius"], info["caustic_radius"]]), g["info_radii"])
    nn = np.tile(np.array([[0, 1, 0]], np.float32), (len(g["probe"]), 1))
    eg = gp.estimate(0, g["probe"], nn)
    assert np.array_equal(eg.view(np.uint32), g["est_global"].view(np.uint32))


def _glass_over_floor():
    """A glass sheet (one interface) that covers a diffuse floor under one lamp: most stored photons are caustic photons."""
    from raytracer_amd.scene import Light, Material, Mesh, SceneData
    sd = SceneData()
    sd.materials = [Material(diffuse=(0.7, 0.7, 0.7), specular=(0.1, 0.1, 0.1), glossiness=20.0),
                    Material(diffuse=(0, 0, 0), specular=(0, 0, 0), refraction=(1.0, 1.0, 1.0), ior=1.5, glossiness=1e4)]
    floor = np.array([[-3, 0, -3], [3, 0, -3], [3, 0, 3], [-3, 0, 3]], np.float32)
    sheet = (floor * np.float32(0.9) + np.array([0, 0.6, 0], np.float32)).astype(np.float32)
    up = np.array([0, 6.0, 0], np.float32)
    # the same pair mirrored above the lamp (a diffuse ceiling behind a second sheet): photons that leave upwards count too
    sd.meshes = [Mesh(floor, [[0, 2, 1], [0, 3, 2]], material=0), Mesh(sheet, [[0, 2, 1], [0, 3, 2]], material=1),
                 Mesh((floor + up).astype(np.float32), [[0, 1, 2], [0, 2, 3]], material=0),
                 Mesh((sheet * np.array([1, -1, 1], np.float32) + up).astype(np.float32), [[0, 1, 2], [0, 2, 3]], material=1)]
    sd.lights = [Light((0.2, 3.0, -0.1), (30.0, 30.0, 28.0), size=0.2)]
    return sd


def test_map_capacity(rtx):
    """PhotonMap::Resize(numPhotons) / Resize(numPhotons / 10) (PhotonMapBuilder.cpp:27-28): a full map takes no more
    photons. Stand-in for the absent cyPhotonMap (parity unpinned): the smallest (photon, bounce) keys stay - on one GPU,
    in the oracle, and when the map is put together from two strided halves."""
    import torch
    sd = _glass_over_floor()
    gs, osc = rtx.Scene(sd), OracleScene(sd)
    N = 20000
    free = OraclePhotons(osc, N, finalize=False).info()          # what emission stores before any limit
    assert free["n_caustic"] > N // 10, "the scene must overflow the caustics map"
    gp, op = rtx.Photons(gs, num_photons=N), OraclePhotons(osc, N)
    gi, oi = gp.info(), op.info()
    assert oi["n_caustic"] == N // 10 and gi["n_caustic"] == N // 10
    assert gi["n_global"] == oi["n_global"] <= N
    a, b = _sorted(gp.download(1)), _sorted(op.download(1))
    assert np.array_equal(a["pos"], b["pos"]) and np.array_equal(a["power"] * (np.float32(1.0) / np.float32(N // 10)), b["power"])
    probe = b["pos"][:200].astype(np.float32)
    nrm = np.tile(np.array([[0, 1, 0]], np.float32), (len(probe), 1))
    e_full = gp.estimate(1, probe, nrm)
    assert np.array_equal(e_full.view(np.uint32), op.estimate(1, probe, nrm).view(np.uint32)) and np.abs(e_full).max() > 0
    # two strided halves, gathered and installed with rtx_photon_set: the same map
    ha = rtx.Photons(gs, num_photons=N, first_photon=0, photon_stride=2)
    hb = rtx.Photons(gs, num_photons=N, first_photon=1, photon_stride=2)
    na, nb = ha.info()["n_caustic"], hb.info()["n_caustic"]
    assert na + nb > N // 10
    buf = torch.empty((na + nb, 12), dtype=torch.float32, device="cuda:0")
    assert ha.copy_raw(1, buf.data_ptr(), na) == na and hb.copy_raw(1, buf[na:].data_ptr(), nb) == nb
    torch.cuda.synchronize()
    ha.set_raw(1, buf.data_ptr(), na + nb, N)
    assert ha.info()["n_caustic"] == N // 10
    assert np.array_equal(ha.estimate(1, probe, nrm).view(np.uint32), e_full.view(np.uint32))
    # renders with the clamped map agree as well
    from raytracer_amd.scene import make_camera
    cam = make_camera((0.5, 2.0, 5.0), (0, 0.3, 0), fov=40, focaldist=5.0, dof=0.0, width=48, height=27)
    rp = render_params(spp_min=4, max_bounces=4, use_photons=1)
    assert_path_samples_identical(gs.render_samples(cam, rp, gp)[0], osc.render_samples(cam, rp, op))
