"""GPU parity of the light-transport path (ray-gen, extend, shade, shadow, resolve, film, adaptive rounds) against
the CPU oracle, through the C ABI. Same sampler sequence and the same counter-based RNG on both sides.

Bars (fp32): per-sample radiance and primary distance BIT-IDENTICAL to the oracle, at every depth - both sides
evaluate the same IEEE operations in the same order, the elementary functions included (raytracer_amd/csrc/rtx_math.h;
with the GPU's own powf / sinf against glibc's, 0.2 % of the samples of this scene parted from the oracle at 64 bounces:
a last-bit difference in the Blinn lobe is a different path a few bounces later). Film sums are float atomics in
scheduling order: accumulated images are compared with a per-pixel tolerance (RMSE <= 1e-5 of the mean radiance).
"""
import numpy as np
import pytest

from raytracer_amd import _ffi, synth
from raytracer_amd.scene import render_params
from tests.helpers import random_rays, assert_path_samples_identical, exact_fraction
from tests.oracle_ffi import OracleScene

pytestmark = pytest.mark.gpu
import os
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def small(rtx):
    sd = synth.build_scene(0.003)
    cam = synth.cerealbar_camera(96, 54)
    return sd, cam, rtx.Scene(sd), OracleScene(sd)


def frac_within(g, o, tol=1e-4):
    d = np.abs(g[..., :3] - o[..., :3]).max(-1)
    return float((d <= tol * (1 + np.abs(o[..., :3]).max(-1))).mean())


BAR = {1: 0.9999, 4: 0.998, 64: 0.995}     # share of samples within 1e-4 * (1 + |ref|), by max_bounces


@pytest.mark.parametrize("max_bounces", [1, 4, 64])
def test_per_sample_radiance(small, max_bounces):
    """max_bounces=1 is config C2 (primary + soft-shadow rays, direct lighting); 64 is the reference's path loop."""
    sd, cam, gs, osc = small
    rp = render_params(spp_min=8, max_bounces=max_bounces)
    g, st = gs.render_samples(cam, rp)
    o = osc.render_samples(cam, rp)
    assert np.isfinite(g).all()
    assert np.array_equal(g[..., 3], o[..., 3])              # primary distance (thin-lens ray generation + first extend)
    assert_path_samples_identical(g, o)
    assert st["rays_primary"] == cam.width * cam.height * 8
    assert (g[..., :3] <= 10.0).all()                      # kRadianceClamp


def test_image_film_z_counts(small):
    sd, cam, gs, osc = small
    rp = render_params(spp_min=16, max_bounces=64)
    g = gs.render(cam, rp)
    o = osc.render(cam, rp)
    rmse = np.sqrt(np.mean((g["linear"] - o["linear"]) ** 2))
    assert rmse <= 1e-5 * o["linear"].mean(), rmse                    # same samples; the film sums them in another order
    assert np.allclose(g["linear"], o["linear"], rtol=2e-5, atol=2e-6)
    assert np.array_equal(g["samples"], o["samples"]) and (g["samples"] == 16).all()
    assert np.array_equal(g["z"], o["z"])
    assert (np.abs(g["rgb8"].astype(int) - o["rgb8"].astype(int)) <= 1).all()     # a value within rounding of a level boundary
    assert (g["rgb8"] == o["rgb8"]).mean() >= 0.999
    # rays actually traced: primary and bounce counts agree with the oracle to within path-divergence noise; the GPU
    # traces no shadow ray for light samples whose BSDF contribution is exactly zero
    assert g["stats"]["rays_primary"] == o["rays"]["primary"]
    assert g["stats"]["rays_bounce"] == o["rays"]["bounce"]              # identical paths: identical ray counts
    assert 0 < g["stats"]["rays_shadow"] <= o["rays"]["shadow"]


def test_filters_and_pass_size_independence(small, rtx):
    """Every reconstruction filter of RayCaster.cpp:643-690; results do not depend on how samples are batched."""
    sd, cam, gs, osc = small
    for ft, rad in ((_ffi.FILTER_GAUSSIAN, 1.5), (_ffi.FILTER_MITCHELL, 0.0), (_ffi.FILTER_LANCZOS, 2.0), (_ffi.FILTER_TENT, 0.0)):
        rp = render_params(spp_min=4, max_bounces=2, filter_type=ft, filter_radius=rad)
        g, o = gs.render(cam, rp), osc.render(cam, rp)
        rmse = np.sqrt(np.mean((g["linear"] - o["linear"]) ** 2))
        assert rmse <= 1e-5 * np.abs(o["linear"]).mean(), (ft, rmse)
    a, _ = gs.render_samples(cam, render_params(spp_min=6, max_bounces=8, samples_per_pass=1))
    b, _ = gs.render_samples(cam, render_params(spp_min=6, max_bounces=8, samples_per_pass=4))
    c, _ = gs.render_samples(cam, render_params(spp_min=6, max_bounces=8, samples_per_pass=6))
    assert np.array_equal(a, b) and np.array_equal(a, c)    # bit-identical per sample


def test_adaptive_sample_maps(small):
    """Adaptive rounds (RayCaster.cpp:843-980): per-pixel sample counts follow the oracle's."""
    sd, cam, gs, osc = small
    rp = render_params(spp_min=8, spp_max=32, max_bounces=8, ci_halfwidth=0.02)
    g, o = gs.render(cam, rp), osc.render(cam, rp)
    assert set(np.unique(g["samples"])) <= {8, 16, 24, 32}
    assert np.array_equal(g["samples"], o["samples"])              # identical samples: identical refinement decisions
    assert (g["samples"] > 8).any() and (g["samples"] == 8).any()
    rmse = np.sqrt(np.mean((g["linear"] - o["linear"]) ** 2))
    assert rmse <= 1e-5 * o["linear"].mean()
    assert np.array_equal(g["z"], o["z"])


def test_visibility_chains(small):
    """Visible()/VisibleSimple() (RayCaster.cpp:178-278) on random segments through the glass teapots and panes."""
    sd, cam, gs, osc = small
    rng = np.random.default_rng(5)
    rays = random_rays(rng, 30000, osc.bounds(), tmax=3.0)
    nrm = rng.normal(size=(30000, 3)).astype(np.float32)
    for ut in (True, False):
        gv, ov = gs.trace_visibility(rays, nrm, ut), osc.trace_visibility(rays, nrm, ut)
        assert np.array_equal(gv, ov)
    part = osc.trace_visibility(rays, nrm, True)
    assert ((part > 0) & (part < 1)).sum() > 20            # transmissive chains are exercised


def test_tile_sharding_sums_to_full_frame(small, rtx):
    """Two 'ranks' on one GPU: films of disjoint bucket sets add up to the unsharded film (what RCCL reduces)."""
    import torch
    sd, cam, gs, osc = small
    n = cam.width * cam.height
    dev = torch.device("cuda:0")

    def film():
        t = [torch.empty((n, 4), device=dev), torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, device=dev),
             torch.empty((n, 4), device=dev), torch.empty((n, 4), device=dev)]
        return t, _ffi.Film(*[x.data_ptr() for x in t])

    films = []
    for rank, world in ((0, 1), (0, 2), (1, 2)):
        t, f = film()
        rtx.film_clear(f, cam.width, cam.height)
        gs.render_accumulate(cam, render_params(spp_min=4, max_bounces=8, rank=rank, world=world), f)
        torch.cuda.synchronize()
        films.append(t)
    full, a, b = films
    assert torch.equal(a[1] + b[1], full[1]) and int(full[1].sum()) == n * 4
    assert torch.equal(torch.minimum(a[2], b[2]), full[2])
    assert torch.allclose(a[0] + b[0], full[0], rtol=1e-4, atol=1e-5)
    res = rtx.film_resolve(_ffi.Film(*[x.data_ptr() for x in full]), cam, want_aov=True)
    o = osc.render(cam, render_params(spp_min=4, max_bounces=8))
    hit = o["albedo"][..., 3] > 0
    assert np.array_equal(res["albedo"][hit], o["albedo"][..., :3][hit])            # first-hit guides (RayCaster.cpp:709-714)
    assert np.array_equal(res["normal"][hit], o["normal"][..., :3][hit])


def test_golden_path_fixture(rtx):
    """The committed oracle fixture (tools/make_golden.py) on the GPU box without running the oracle."""
    g = np.load(os.path.join(GOLD, "path_small.npz"))
    sd = synth.build_scene(0.002)
    cam = synth.cerealbar_camera(48, 27)
    gs = rtx.Scene(sd)
    s, _ = gs.render_samples(cam, render_params(spp_min=4, max_bounces=64))
    assert_path_samples_identical(s, g["samples"])
    ada = gs.render(cam, render_params(spp_min=8, spp_max=32, max_bounces=8, ci_halfwidth=0.02))
    assert np.array_equal(ada["samples"], g["adaptive_samples"])


def test_error_paths(small, rtx):
    sd, cam, gs, osc = small
    with pytest.raises(rtx.RtxError, match="spp"):
        gs.render(cam, render_params(spp_min=0))
    with pytest.raises(rtx.RtxError, match="rank"):
        gs.render(cam, render_params(spp_min=1, rank=3, world=2))
    big = synth.cerealbar_camera(4096, 4096)
    with pytest.raises(rtx.RtxError, match="overflows"):
        gs.render(big, render_params(spp_min=1))             # sampleIndex would overflow int (RayCaster.cpp:784)


def test_scheduling_switches_do_not_change_results(small, monkeypatch):
    """The shadow kernel's early exit, the side stream and the tail kernel are scheduling: per-sample radiance with each
    of them switched off is bit-identical (shade_kernel and tail_kernel instantiate the same shading body, and every
    translation unit is built with IEEE division and square root)."""
    sd, cam, gs, osc = small
    rp = render_params(spp_min=4, max_bounces=64)
    base, st0 = gs.render_samples(cam, rp)
    monkeypatch.setenv("RTX_AMD_SHADOW_EARLY_EXIT", "0")
    full, st1 = gs.render_samples(cam, rp)
    monkeypatch.delenv("RTX_AMD_SHADOW_EARLY_EXIT")
    assert np.array_equal(base, full)
    assert st1["rays_shadow"] == st0["rays_shadow"]                 # the same chains hop for hop: the exit shortens traversals only
    monkeypatch.setenv("RTX_AMD_OVERLAP", "0")
    serial, _ = gs.render_samples(cam, rp)
    monkeypatch.delenv("RTX_AMD_OVERLAP")
    assert np.array_equal(base, serial)
    monkeypatch.setenv("RTX_AMD_TAIL_PATHS", "0")
    loop, _ = gs.render_samples(cam, rp)
    monkeypatch.delenv("RTX_AMD_TAIL_PATHS")
    assert np.array_equal(loop, base)
    # compaction of the path state inside a pass: forced at the first bounce of this small pass and at every
    # bounce that finds a quarter of the last count; the paths then end in the bounce loop (tail kernel off), in the tail
    # kernel, or are never moved (switched off)
    assert st0["repacks"] == 0                                      # (a pass this small is not moved by default)
    monkeypatch.setenv("RTX_AMD_REPACK_DIV", "1")
    monkeypatch.setenv("RTX_AMD_REPACK_MIN", "0")
    moved_tail, st2 = gs.render_samples(cam, rp)
    monkeypatch.setenv("RTX_AMD_TAIL_PATHS", "0")
    moved, st3 = gs.render_samples(cam, rp)
    monkeypatch.delenv("RTX_AMD_TAIL_PATHS")
    monkeypatch.delenv("RTX_AMD_REPACK_DIV")
    monkeypatch.delenv("RTX_AMD_REPACK_MIN")
    assert st2["repacks"] >= 1 and st3["repacks"] >= 3
    assert np.array_equal(moved_tail, base) and np.array_equal(moved, base)
    assert st3["rays_shadow"] == st0["rays_shadow"] and st3["rays_bounce"] == st0["rays_bounce"]
    monkeypatch.setenv("RTX_AMD_REPACK", "0")
    fixed, st4 = gs.render_samples(cam, rp)
    monkeypatch.delenv("RTX_AMD_REPACK")
    assert st4["repacks"] == 0 and np.array_equal(fixed, base)
    # the persistent kernels' walk of their queue (segqueue.h SegWalk): every extend / shadow launch as one front, or none
    monkeypatch.setenv("RTX_AMD_FRONT_MIN", "0")
    monkeypatch.setenv("RTX_AMD_FRONT_MAX", "4000000000")
    front, st5 = gs.render_samples(cam, rp)
    monkeypatch.setenv("RTX_AMD_FRONT_MAX", "0")
    by_segment, st6 = gs.render_samples(cam, rp)
    monkeypatch.delenv("RTX_AMD_FRONT_MIN")
    monkeypatch.delenv("RTX_AMD_FRONT_MAX")
    assert np.array_equal(front, base) and np.array_equal(by_segment, base)
    assert st5["rays_shadow"] == st0["rays_shadow"] and st6["rays_bounce"] == st0["rays_bounce"]
    # the path pool as an allocation per array (the fallback of ensure_pool) instead of one slab: a scene of its own, since a
    # scene keeps its pool
    monkeypatch.setenv("RTX_AMD_POOL_SLAB", "0")
    apart, _ = type(gs)(sd).render_samples(cam, rp)
    monkeypatch.delenv("RTX_AMD_POOL_SLAB")
    assert np.array_equal(apart, base)


@pytest.mark.parametrize("mode", [_ffi.SHADING_PATH, _ffi.SHADING_LEGACY])
def test_textured_glossiness(rtx, mode):
    """Glossiness() is a TexturedFloat in the reference (Shading.cpp:238, Shading_Legacy.cpp:375: sInfo.Eval(Glossiness())):
    rtx_material_desc::glossiness_tex scales the value by the map's red channel, in both shading modes."""
    sd = synth.build_scene(0.003)
    cam = synth.cerealbar_camera(96, 54)
    rp = render_params(spp_min=4, max_bounces=4, shading_mode=mode)
    plain, _ = rtx.Scene(sd).render_samples(cam, rp)
    for m in sd.materials:
        if m.glossiness > 50.0 and m.glossiness < 1e4:
            m.glossiness_tex = 0                              # the value-noise map the cereal materials use for bump
    gs, osc = rtx.Scene(sd), OracleScene(sd)
    g, _ = gs.render_samples(cam, rp)
    o = osc.render_samples(cam, rp)
    assert_path_samples_identical(g, o)                          # both modes: the oracle's bits
    assert np.abs(g[..., :3] - plain[..., :3]).mean() > 1e-4   # the map does change the highlights
