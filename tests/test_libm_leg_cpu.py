"""The oracle's own two legs against each other: the build whose elementary functions are the routines shared with the
kernels (raytracer_amd/csrc/rtx_math.h) and the build that calls the platform libm as the reference does. If the shared
routines were wrong in any way that matters to an image, this is where it would show - the GPU parity tests compare the
kernels with the first leg only."""
import numpy as np

from tests import libm_leg
from tests.oracle_ffi import OracleScene


def test_shared_math_oracle_renders_the_image_of_the_platform_libm_oracle():
    sd, cam = libm_leg.scene()
    shared, platform = OracleScene(sd), OracleScene(sd, libm=True)
    img_shared = shared.render(cam, libm_leg.params(), nthreads=8)["linear"]
    img_platform = platform.render(cam, libm_leg.params(), nthreads=8)["linear"]
    img_other_seed = platform.render(cam, libm_leg.params(libm_leg.SEED_B), nthreads=8)["linear"]
    assert np.isfinite(img_shared).all() and np.isfinite(img_platform).all()
    noise = libm_leg.rmse(img_other_seed, img_platform)        # what the renderer's own randomness does to the image at this spp
    diff = libm_leg.rmse(img_shared, img_platform)
    assert noise > 0.0
    assert diff <= 1.5 * noise, (diff, noise)
    # and far below it in fact: most samples take the same path with either libm
    same = (img_shared.view(np.uint32) == img_platform.view(np.uint32)).all(-1).mean()
    print(f"rmse shared vs platform libm {diff:.3e}, seed-to-seed {noise:.3e}, pixels bit-identical {same:.3f}")
    assert diff <= 0.5 * noise, (diff, noise)


def test_the_two_legs_agree_on_the_elementary_functions_to_a_few_ulp():
    """rtx_math.h against glibc on the path's argument ranges: <= 4 ulp, special values alike."""
    import ctypes as C
    from tests.math_inputs import FN, inputs, ulp_error
    from tests.oracle_ffi import lib
    for name in FN:
        x, y, n = inputs(name, 100000, seed=5)
        outs = []
        for libm in (False, True):
            out = np.empty_like(x)
            lib(libm).orc_math(FN[name], C.c_void_p(x.ctypes.data), C.c_void_p(y.ctypes.data), C.c_size_t(len(x)), C.c_void_p(out.ctypes.data))
            outs.append(out)
        a, b = outs[0][:n], outs[1][:n]
        fin = np.isfinite(b) & (np.abs(b) > 1e-37) & (np.abs(b) < 3e38)
        if name == "tanf":
            fin &= np.abs(b) < 1e4
        if name in ("sinf", "cosf", "tanf"):
            fin &= np.abs(x[:n]) <= 8192
        err = ulp_error(a[fin], b[fin].astype(np.float64))
        assert err.max() <= 4.0, (name, err.max())
