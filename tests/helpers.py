"""Shared helpers for the parity tests."""
import numpy as np

from raytracer_amd import _ffi


def random_rays(rng, n, bounds, tmax=_ffi.BIGFLOAT):
    """Rays from points around the scene box towards points inside it."""
    ray_dt, _, _ = _ffi.np_dtypes()
    lo, hi = np.array(bounds[:3], np.float32), np.array(bounds[3:], np.float32)
    c, e = 0.5 * (lo + hi), (hi - lo)
    o = (c + (rng.random((n, 3), dtype=np.float32) - 0.5) * e * 3.0).astype(np.float32)
    tgt = (lo + rng.random((n, 3), dtype=np.float32) * e).astype(np.float32)
    d = tgt - o
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    rays = np.zeros(n, dtype=ray_dt)
    rays["o"], rays["d"], rays["tmax"] = o, d, tmax
    return rays


def global_prim(hits, face_base):
    fb = np.asarray(face_base)
    return np.where(hits["obj"] >= 0, fb[np.maximum(hits["obj"], 0)] + hits["face"], -1)


def compare_hits(g, o, face_base, id_tol=1e-4, t_rel=1e-5):
    """north_star bar: ids agree on >= 99.99 % of rays, |dt|/t <= 1e-5 where they agree."""
    gp, op = global_prim(g, face_base), global_prim(o, face_base)
    same = (gp == op) & (g["light"] == o["light"])
    frac = 1.0 - same.mean()
    assert frac <= id_tol, f"hit id mismatch on {frac:.3e} of rays"
    hit = same & ((gp >= 0) | (g["light"] >= 0))
    if hit.any():
        rel = np.abs(g["t"][hit] - o["t"][hit]) / np.maximum(np.abs(o["t"][hit]), 1e-20)
        assert rel.max() <= t_rel, f"max relative t error {rel.max():.3e}"
    return same, hit


# ---- per-sample radiance against the oracle ---------------------------------------------------------------------------
# Kernels and oracle evaluate the same IEEE operations in the same order, elementary functions included
# (raytracer_amd/csrc/rtx_math.h), so in path mode a sample is either bit-identical or wrong. The legacy mode's kernel
# sums the linearised recursion tree of a sample in another order than the recursive oracle: float summation order only.

def exact_fraction(g, o):
    """Share of samples whose r, g, b are bit-identical."""
    a = np.ascontiguousarray(g[..., :3], np.float32).view(np.uint32)
    b = np.ascontiguousarray(o[..., :3], np.float32).view(np.uint32)
    return float((a == b).all(-1).mean())


def within_fraction(g, o, tol):
    d = np.abs(g[..., :3] - o[..., :3]).max(-1)
    return float((d <= tol * (1 + np.abs(o[..., :3]).max(-1))).mean())


def assert_path_samples_identical(g, o, min_exact=1.0):
    f = exact_fraction(g, o)
    assert f >= min_exact, f"only {f:.6f} of the samples are bit-identical to the oracle (within 1e-6: {within_fraction(g, o, 1e-6):.6f})"


def assert_legacy_samples_close(g, o, tol=1e-5, min_frac=0.9999):
    """Legacy-mode samples: bit-identical as well since legacy_kernel keeps the recursion's association (the name is
    round 1's, when the kernel summed a sample's tree in another order)."""
    assert_path_samples_identical(g, o)
