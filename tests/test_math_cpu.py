"""The elementary functions shared by the kernels and the oracle (raytracer_amd/csrc/rtx_math.h) against float64: they
stand in for the platform libm the reference calls, so they have to be as good as one. Also: the host side of the
product library and the oracle (two compilers, one source) agree bit for bit."""
import ctypes as C

import numpy as np
import pytest

from raytracer_amd import _ffi
from tests.math_inputs import FN, inputs, reference64, ulp_error
from tests.oracle_ffi import lib as oracle_lib

BAR = dict(powf=1.0, expf=1.0, sinf=2.0, cosf=2.0, tanf=4.0, asinf=3.0, acosf=3.0, atan2f=3.0)   # ulp, on the path's ranges


def oracle_eval(name, x, y):
    out = np.empty_like(x)
    oracle_lib().orc_math(FN[name], C.c_void_p(x.ctypes.data), C.c_void_p(y.ctypes.data), C.c_size_t(len(x)), C.c_void_p(out.ctypes.data))
    return out


@pytest.mark.parametrize("name", list(FN))
def test_accuracy_against_float64(name):
    x, y, n = inputs(name, 200000, seed=1)
    got = oracle_eval(name, x, y)
    ref = reference64(name, x, y)
    fin = np.isfinite(ref[:n]) & (np.abs(ref[:n]) > 1e-37) & (np.abs(ref[:n]) < 3e38)
    if name == "tanf":
        fin &= np.abs(ref[:n]) < 1e4            # next to a pole the quotient amplifies the error of the cosine
    err = ulp_error(got[:n][fin], ref[:n][fin])
    assert err.max() <= BAR[name], (name, err.max(), x[:n][fin][err.argmax()], y[:n][fin][err.argmax()])
    # special values: NaN where the exact function is NaN, infinities and zeros where it overflows / underflows
    gs, rs = got[n:], ref[n:].astype(np.float32)
    if name in ("sinf", "cosf", "tanf"):
        ok = np.abs(x[n:]) <= 8192               # beyond the stated range only NaN-ness is compared
        assert np.array_equal(np.isnan(gs), np.isnan(rs))
    else:
        ok = np.ones(len(gs), bool)
    same = (np.isnan(gs) & np.isnan(rs)) | (gs == rs) | (np.abs(gs.astype(np.float64) - ref[n:]) <= 4 * np.spacing(np.abs(rs)).astype(np.float64))
    assert same[ok].all(), (name, x[n:][ok][~same[ok]][:5], y[n:][ok][~same[ok]][:5], gs[ok][~same[ok]][:5], rs[ok][~same[ok]][:5])


@pytest.mark.parametrize("name", list(FN))
def test_host_side_of_the_library_equals_the_oracle_bit_for_bit(name):
    x, y, n = inputs(name, 100000, seed=2)
    lib = _ffi.load_library()
    out = np.empty_like(x)
    assert lib.rtx_debug_math(FN[name], x.ctypes.data, y.ctypes.data, len(x), out.ctypes.data, 0) == 0
    assert np.array_equal(out.view(np.uint32), oracle_eval(name, x, y).view(np.uint32))


def test_unorm8_is_the_ieee_division_for_every_texel_value():
    """rtx_unorm8(x) (three instructions in a kernel) must be the float that (float)x / 255.0f gives - the reference's and the
    oracle's texel conversion - for each of the 256 values a texel can hold; checked on the host side of the library here
    and on the device in tests/test_math_gpu.py."""
    x = np.arange(256, dtype=np.float32)
    lib = _ffi.load_library()
    out = np.empty_like(x)
    assert lib.rtx_debug_math(8, x.ctypes.data, x.ctypes.data, len(x), out.ctypes.data, 0) == 0
    assert np.array_equal(out.view(np.uint32), (x / np.float32(255.0)).view(np.uint32))
