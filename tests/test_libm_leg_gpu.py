"""The GPU render against the oracle leg that shares no arithmetic source with it (platform libm, as the reference calls it):
north_star's "rendered images must match within a stated per-pixel RMSE at equal spp", stated against the reference's own
libm. Bar: RMSE(GPU, oracle) <= 1.5 x RMSE(oracle seed A, oracle seed B) at 64 spp."""
import numpy as np
import pytest

from tests import libm_leg
from tests.oracle_ffi import OracleScene

pytestmark = pytest.mark.gpu


def test_gpu_image_against_the_platform_libm_oracle(rtx):
    sd, cam = libm_leg.scene()
    gs, platform = rtx.Scene(sd), OracleScene(sd, libm=True)
    gpu = gs.render(cam, libm_leg.params())["linear"]
    ref = platform.render(cam, libm_leg.params(), nthreads=8)["linear"]
    ref_other_seed = platform.render(cam, libm_leg.params(libm_leg.SEED_B), nthreads=8)["linear"]
    assert np.isfinite(gpu).all()
    noise = libm_leg.rmse(ref_other_seed, ref)
    diff = libm_leg.rmse(gpu.reshape(ref.shape), ref)
    print(f"rmse GPU vs platform-libm oracle {diff:.3e}, oracle seed-to-seed {noise:.3e}")
    assert noise > 0.0
    assert diff <= 1.5 * noise, (diff, noise)
