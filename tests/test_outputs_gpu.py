"""Outputs after the path: the 3x3 median denoiser of the reference (Denoiser.cpp:19-134) as a HIP
kernel, against its oracle restatement."""
import numpy as np
import pytest

from tests import oracle_ffi

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(1, 1), (2, 3), (37, 53), (270, 480)])
def test_median_denoise_matches_oracle(rtx, shape):
    from raytracer_amd import api
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    img = rng.gamma(0.7, 0.6, size=shape + (3,)).astype(np.float32)      # values above 1 and fireflies
    img[rng.random(shape) < 0.03] = 25.0
    img[rng.random(shape) < 0.02] = 0.0
    for srgb in (False, True):
        gl, g8 = api.median_denoise(img, srgb)
        ol, o8 = oracle_ffi.median_denoise(img, srgb)
        assert np.array_equal(gl, ol)                                    # a median selects: bit-exact
        d = np.abs(g8.astype(np.int32) - o8.astype(np.int32))
        assert d.max() <= (1 if srgb else 0)                             # powf in the sRGB curve: one level at most
        assert (d > 0).mean() < 0.01
    # fireflies are gone, constant regions are kept
    flat = np.full((16, 16, 3), 0.25, np.float32)
    flat[8, 8] = 100.0
    out, _ = api.median_denoise(flat)
    assert np.array_equal(out, np.full_like(flat, 0.25))
