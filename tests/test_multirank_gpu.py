"""Two ranks of the HIP path against one rank: (1) render_accumulate + reduce_film over consecutive
frames equals the single-rank film (counts and z exact, colour up to float summation order, guides exact) - this is the
loop bench.py times; (2) the photon map all-gathered from two ranks that traced half of the ids each equals the
single-rank map: stored counts, irradiance estimates and the caustic term in the path loop bit-identical.
The ranks share the test GPU and use gloo (the nccl branch needs one GPU per rank: RTX_TEST_BACKEND=nccl)."""
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _spawn(world, tmp):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = str(s.getsockname()[1])
    s.close()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "multirank_worker.py"), str(r), str(world), port, tmp],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=280)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-3000:] for o in outs)
    return [dict(np.load(os.path.join(tmp, f"rank{r}_of{world}.npz"))) for r in range(world)]


@pytest.fixture(scope="module")
def runs(rtx):
    with tempfile.TemporaryDirectory() as tmp:
        one = _spawn(1, tmp)[0]
        two = _spawn(2, tmp)
    return one, two


def test_consecutive_frames_reduce_to_the_single_rank_film(runs):
    one, (r0, r1) = runs
    for step in range(3):
        assert np.array_equal(r0[f"count{step}"], one[f"count{step}"]) and (one[f"count{step}"] == 4).all()
        assert np.array_equal(r0[f"z{step}"], one[f"z{step}"])
        assert np.allclose(r0[f"accum{step}"], one[f"accum{step}"], rtol=1e-4, atol=1e-5)
        assert np.array_equal(r0[f"albedo{step}"], one[f"albedo{step}"])
        assert np.array_equal(r0[f"normal{step}"], one[f"normal{step}"])
    # the steps differ (different seeds): a stale or half-cleared film would show up as a step repeating
    assert not np.allclose(one["accum0"], one["accum1"])


def test_all_gathered_photon_map_equals_single_rank_map(runs):
    one, (r0, r1) = runs
    assert (r0["mine"] + r1["mine"] == one["total"]).all()             # ids are split, nothing lost or doubled
    assert (r0["total"] == one["total"]).all() and (r1["total"] == one["total"]).all()
    assert (r0["mine"] > 0).all() and (r1["mine"] > 0).all()
    assert np.array_equal(r0["probes"], one["probes"]) and np.array_equal(r1["probes"], one["probes"])
    for k in ("est0", "est1"):
        ref = one[k]
        assert np.abs(ref).max() > 0
        for r in (r0, r1):
            assert np.array_equal(r[k].view(np.uint32), ref.view(np.uint32))   # maps are stored in (cell, photon key) order
    for r in (r0, r1):
        assert np.array_equal(r["caustic_samples"].view(np.uint32), one["caustic_samples"].view(np.uint32))


def test_rccl_code_path_with_one_rank(runs):
    """The box has one GPU, and RCCL refuses two ranks on one device: so the nccl branches of reduce_film and
    all_gather_photons (collectives on device tensors, all_gather_into_tensor, the host synchronisation against the
    library's own HIP runtime) run here in a group of ONE rank whose collectives are forced, and must change nothing."""
    one, _ = runs
    env = dict(os.environ, RTX_TEST_BACKEND="nccl", RTX_AMD_DIST_FORCE="1")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = str(s.getsockname()[1])
    s.close()
    with tempfile.TemporaryDirectory() as tmp:
        p = subprocess.run([sys.executable, os.path.join(HERE, "multirank_worker.py"), "0", "1", port, tmp], env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=280)
        assert p.returncode == 0, p.stdout[-3000:]
        got = dict(np.load(os.path.join(tmp, "rank0_of1_forced.npz")))
    for k in ("count0", "z0", "albedo2", "normal2", "probes", "total", "est0", "est1", "caustic_samples"):
        assert np.array_equal(got[k], one[k]), k
    assert np.allclose(got["accum1"], one["accum1"], rtol=1e-4, atol=1e-5)


# ---- several members in ONE process (rtx_group: the C++ host's multi-GPU path) ----------------------------------
# The test box has one GPU: members share it (device ids repeat), which exercises everything but the peer copies
# themselves (a member on the root's device is merged in place).

@pytest.fixture(scope="module")
def group_env(rtx):
    from raytracer_amd import synth
    sd = synth.build_scene(0.003)
    return sd, rtx.Scene(sd), synth.cerealbar_camera(96, 54)


@pytest.mark.parametrize("members", [1, 2, 3])
def test_group_render_equals_single_device(group_env, rtx, members, monkeypatch):
    from raytracer_amd.scene import render_params
    sd, gs, cam = group_env
    grp = rtx.Group(sd, [0] * members)
    rp = render_params(spp_min=4, max_bounces=8)
    ref = gs.render(cam, rp)
    got = grp.render(cam, rp)
    assert np.array_equal(got["samples"], ref["samples"]) and (got["samples"] == 4).all()
    assert np.array_equal(got["z"], ref["z"])
    assert np.allclose(got["linear"], ref["linear"], rtol=2e-4, atol=2e-5)          # float summation order only
    for k in ("rays_primary", "rays_bounce", "rays_shadow", "paths"):
        assert got["stats"][k] == ref["stats"][k]
    # guides: first-hit albedo / normal of every pixel, whichever member owns it
    import torch
    from raytracer_amd import _ffi
    n = cam.width * cam.height
    t = [torch.empty((n, 4), device="cuda:0"), torch.empty(n, dtype=torch.int32, device="cuda:0"), torch.empty(n, device="cuda:0"),
         torch.empty((n, 4), device="cuda:0"), torch.empty((n, 4), device="cuda:0")]
    f = _ffi.Film(*[x.data_ptr() for x in t])
    rtx.film_clear(f, cam.width, cam.height)
    gs.render_accumulate(cam, rp, f)
    one = rtx.film_resolve(f, cam, want_aov=True)
    assert np.array_equal(got["albedo"], one["albedo"]) and np.array_equal(got["normal"], one["normal"])
    # adaptive sampling: after every round the unconverged buckets are dealt out again over the members (statistics
    # collected on member 0, greedy LPT, merged statistics back: rtx_group.cpp). A pixel's decision uses only its own
    # samples, whoever traced them, so the sample maps are those of one GPU - and of the static split.
    rpa = render_params(spp_min=8, spp_max=40, max_bounces=8, ci_halfwidth=0.02)
    a, b = grp.render(cam, rpa), gs.render(cam, rpa)
    assert np.array_equal(a["samples"], b["samples"]) and (a["samples"] > 16).any() and (a["samples"] == 8).any()
    assert np.array_equal(a["z"], b["z"])
    assert np.allclose(a["linear"], b["linear"], rtol=2e-4, atol=2e-5)
    for k in ("rays_primary", "rays_bounce", "rays_shadow", "paths"):
        assert a["stats"][k] == b["stats"][k], k
    monkeypatch.setenv("RTX_AMD_GROUP_STATIC", "1")
    c = grp.render(cam, rpa)
    monkeypatch.delenv("RTX_AMD_GROUP_STATIC")
    assert np.array_equal(c["samples"], b["samples"])
    assert np.allclose(c["linear"], b["linear"], rtol=2e-4, atol=2e-5)
    grp.close()


def test_group_photon_map_equals_single_device(group_env, rtx):
    from raytracer_amd.scene import render_params
    from raytracer_amd import synth
    sd, gs, cam = group_env
    N = 200000
    single = rtx.Photons(gs, num_photons=N)
    grp = rtx.Group(sd, [0, 0])
    with pytest.raises(rtx.RtxError, match="no photon map"):
        grp.member_photons(0)
    st = grp.photon_build(num_photons=N)
    assert st["rays_photon"] == single.stats["rays_photon"]
    pos = single.download(1)["pos"]
    probe = (pos[:: max(1, len(pos) // 200)][:200] + np.float32(0.002)).astype(np.float32)
    nrm = np.tile(np.array([[0, 1, 0]], np.float32), (len(probe), 1))
    for member in (0, 1):
        ph = grp.member_photons(member)
        assert ph.info() == single.info()
        for which in (0, 1):
            ref = single.estimate(which, probe, nrm)
            assert np.abs(ref).max() > 0
            assert np.array_equal(ph.estimate(which, probe, nrm).view(np.uint32), ref.view(np.uint32))
    small = synth.cerealbar_camera(48, 27)
    rp = render_params(spp_min=4, max_bounces=4, use_photons=1)
    a, b = grp.render(small, rp), gs.render(small, rp, single)
    assert np.allclose(a["linear"], b["linear"], rtol=1e-3, atol=1e-5)
    dark = gs.render(small, render_params(spp_min=4, max_bounces=4))
    assert np.abs(b["linear"] - dark["linear"]).max() > 1e-4                        # the caustic map does contribute
    grp.close()


def test_group_rejects_bad_arguments(group_env, rtx):
    sd, gs, cam = group_env
    with pytest.raises(rtx.RtxError, match="out of range"):
        rtx.Group(sd, [0, 99])
    with pytest.raises(rtx.RtxError, match="RTX_GROUP_MAX"):
        rtx.Group(sd, [0] * 17)
