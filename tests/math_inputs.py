"""Inputs for the tests of the shared elementary functions (raytracer_amd/csrc/rtx_math.h)."""
import numpy as np

FN = dict(powf=0, expf=1, sinf=2, cosf=3, tanf=4, asinf=5, acosf=6, atan2f=7)
SPECIAL = np.array([0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 2.0, -2.0, 3.0, 1e-45, -1e-45, 1.17549435e-38, 3.4028235e38, -3.4028235e38,
                    np.inf, -np.inf, np.nan, 1e5, 0.99999994, 1.0000001, 88.7, -103.9, 1e-20, 16777216.0, 16777217.0, 8388609.0],
                   dtype=np.float32)


def inputs(name, n, seed):
    """(x, y) float32 arrays of the ranges the light-transport path feeds the function, followed by every pair of SPECIAL."""
    g = np.random.default_rng(seed)
    u = lambda lo, hi: g.uniform(lo, hi, n).astype(np.float32)
    if name == "powf":
        # Blinn lobes: cosines to exponents up to 1e5; Schlick: (1 - cos)^5; sRGB: c^0.41666; lobe sampling: u^(1/(alpha+1))
        x = np.concatenate([u(0, 1), 1 - 10 ** u(-7, -1), u(0, 1), u(0, 10), u(0, 1), 10 ** u(-30, 30)])
        y = np.concatenate([10 ** u(0, 5), 10 ** u(2, 5), np.full(n, 5, np.float32), np.full(n, 0.41666, np.float32), 1 / (1 + 10 ** u(0, 5)), u(-3, 3)])
    elif name == "expf":
        x, y = np.concatenate([-u(0, 30), u(-100, 88), u(-1e-3, 1e-3)]), None
    elif name in ("sinf", "cosf", "tanf"):
        x, y = np.concatenate([u(0, 6.2831855), u(-3.1415927, 3.1415927), u(-100, 100), u(-8192, 8192)]), None
    elif name in ("asinf", "acosf"):
        x, y = np.concatenate([u(-1, 1), 1 - 10 ** u(-7, 0), -1 + 10 ** u(-7, 0)]), None
    else:
        x, y = np.concatenate([u(-1, 1), u(-1e3, 1e3), 10 ** u(-20, 20)]), np.concatenate([u(-1, 1), u(-1e-3, 1e-3), -10 ** u(-20, 20)])
    if y is None:
        y = np.zeros_like(x)
    sx, sy = np.meshgrid(SPECIAL, SPECIAL)
    return (np.concatenate([x.astype(np.float32), sx.ravel()]).astype(np.float32), np.concatenate([y.astype(np.float32), sy.ravel()]).astype(np.float32),
            len(x))


def reference64(name, x, y):
    x, y = x.astype(np.float64), y.astype(np.float64)
    with np.errstate(all="ignore"):
        return dict(powf=lambda: np.power(x, y), expf=lambda: np.exp(x), sinf=lambda: np.sin(x), cosf=lambda: np.cos(x), tanf=lambda: np.tan(x),
                    asinf=lambda: np.arcsin(x), acosf=lambda: np.arccos(x), atan2f=lambda: np.arctan2(x, y))[name]()


def ulp_error(got, ref64):
    """Error of float32 results in units of the last place of the exact value."""
    ref32 = ref64.astype(np.float32)
    ulp = np.spacing(np.abs(ref32)).astype(np.float64)
    return np.abs(got.astype(np.float64) - ref64) / ulp
