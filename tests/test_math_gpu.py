"""The shared elementary functions (raytracer_amd/csrc/rtx_math.h) evaluated inside a kernel against the CPU oracle's
evaluation of the same source: every result bit for bit, special values included. This is what makes the per-sample
radiance of the two implementations identical (tests/test_path_gpu.py)."""
import ctypes as C

import numpy as np
import pytest

from tests.math_inputs import FN, inputs
from tests.test_math_cpu import oracle_eval

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", list(FN))
def test_device_equals_oracle_bit_for_bit(rtx, name):
    x, y, n = inputs(name, 1000000, seed=3)
    lib = rtx._ffi.load_library()
    out = np.empty_like(x)
    assert lib.rtx_debug_math(FN[name], x.ctypes.data, y.ctypes.data, len(x), out.ctypes.data, 1) == 0, lib.rtx_last_error()
    ref = oracle_eval(name, x, y)
    bad = out.view(np.uint32) != ref.view(np.uint32)
    assert not bad.any(), (name, int(bad.sum()), x[bad][:5], y[bad][:5], out[bad][:5], ref[bad][:5])


def test_unorm8_on_the_device_is_the_ieee_division(rtx):
    """The texel conversion of the kernels (rtx_unorm8: product with the rounded reciprocal plus one residual step) against
    numpy's float32 division for all 256 texel values."""
    x = np.arange(256, dtype=np.float32)
    lib = rtx._ffi.load_library()
    out = np.empty_like(x)
    assert lib.rtx_debug_math(8, x.ctypes.data, x.ctypes.data, len(x), out.ctypes.data, 1) == 0, lib.rtx_last_error()
    assert np.array_equal(out.view(np.uint32), (x / np.float32(255.0)).view(np.uint32))
