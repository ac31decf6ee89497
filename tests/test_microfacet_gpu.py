"""MtlMicrofacet (GGX + Schlick + delta transmission, reference RayTracer/Shading.cpp:741-936) in the path tracer and
the photon pass, against the CPU oracle."""
import numpy as np
import pytest

from raytracer_amd import _ffi, synth
from raytracer_amd.scene import SceneData, Mesh, Material, Light, make_camera, render_params
from tests.helpers import assert_path_samples_identical
from tests.oracle_ffi import OracleScene, OraclePhotons

pytestmark = pytest.mark.gpu


def micro(name, base, rough, metal, trans=(0, 0, 0), ior=1.5):
    # slots as documented at RTX_MTL_MICROFACET (include/rtx.h)
    return Material(name, type=_ffi.MTL_MICROFACET, diffuse=base, specular=(rough, 0, 0), reflection=(metal, 0, 0),
                    refraction=trans, ior=ior)


@pytest.fixture(scope="module")
def scene(rtx):
    V, N, UV, F = synth.torus(1.0, 0.35, 32, 16)
    sd = SceneData()
    sd.materials = [micro("plastic", (0.7, 0.2, 0.2), 0.4, 0.0), micro("gold", (1.0, 0.77, 0.34), 0.25, 1.0),
                    micro("frosted", (0.1, 0.1, 0.1), 0.1, 0.0, trans=(0.9, 0.9, 0.9)),
                    Material("floor", diffuse=(0.6, 0.6, 0.6), specular=(0.1, 0.1, 0.1), glossiness=20.0)]
    def xf(t, s=1.0):
        return np.array([s, 0, 0, t[0], 0, s, 0, t[1], 0, 0, s, t[2]], np.float32)
    sd.meshes = [Mesh(V, F, vn=N, vt=UV, material=0, xform=xf((-2.2, 0, 0.4))), Mesh(V, F, vn=N, vt=UV, material=1, xform=xf((0, 0, 0.4))),
                 Mesh.sphere(xf((2.2, 0, 0.7), 0.7), material=2), Mesh.plane(xf((0, 0, 0), 6.0), material=3)]
    sd.lights = [Light((1.0, -3.0, 5.0), (40.0, 38.0, 35.0), 0.6, 1.0), Light((-4.0, 2.0, 3.0), (10.0, 12.0, 15.0), 0.0, 1.0)]
    sd.environment = (0.25, 0.3, 0.35)
    cam = make_camera(pos=(0.0, -7.0, 3.0), target=(0.0, 0.0, 0.4), up=(0, 0, 1), fov=38.0, width=72, height=40)
    return sd, cam, rtx.Scene(sd), OracleScene(sd)


@pytest.mark.parametrize("max_bounces", [1, 6])
def test_per_sample_radiance(scene, max_bounces):
    sd, cam, gs, osc = scene
    rp = render_params(spp_min=8, max_bounces=max_bounces)
    g, _ = gs.render_samples(cam, rp)
    o = osc.render_samples(cam, rp)
    assert np.isfinite(g).all()
    assert_path_samples_identical(g, o)
    assert o[..., :3].mean() > 0.02
    assert np.array_equal(g[..., 3], o[..., 3])


def test_photons_through_microfacet_glass(scene, rtx):
    sd, cam, gs, osc = scene
    gp = rtx.Photons(gs, num_photons=40000, max_bounces=2)
    op = OraclePhotons(osc, 40000, max_bounces=2)
    op.finalize()
    gi, oi = gp.info(), op.info()
    for k in ("n_global", "n_caustic"):
        assert abs(gi[k] - oi[k]) <= max(3, 0.005 * oi[k]), (k, gi[k], oi[k])
    assert oi["n_caustic"] > 20 and oi["n_global"] > 20
