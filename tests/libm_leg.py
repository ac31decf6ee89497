"""Image-level comparison against the oracle build that shares no arithmetic source with the product: liboracle_libm.so,
whose powf / expf / sinf / cosf / tanf / asinf / acosf / atan2f are the platform libm, called as the reference calls them
(oracle/oracle_math.h, ORC_PLATFORM_LIBM). Used by tests/test_libm_leg_cpu.py and tests/test_libm_leg_gpu.py.

A last-bit difference in one powf sends a path elsewhere a few bounces later, so samples are not compared one by one: the
criterion is SURVEY.md 8(c)'s - the image under test may differ from the reference render by no more than 1.5 x what two
reference renders with different random streams differ by (the reference itself is stochastic from run to run, F8)."""
import numpy as np

from raytracer_amd import synth
from raytracer_amd.scene import render_params

SCALE, W, H, SPP, BOUNCES = 0.003, 64, 36, 64, 64
SEED_A, SEED_B = 0x9E3779B97F4A7C15, 0x2545F4914F6CDD1D


def scene():
    return synth.build_scene(SCALE), synth.cerealbar_camera(W, H)


def params(seed=SEED_A):
    return render_params(spp_min=SPP, max_bounces=BOUNCES, seed=seed)


def rmse(a, b):
    """per-pixel RMSE over linear RGB clamped to [0, 10] (the radiance clamp), relative to the mean of b"""
    a = np.clip(np.asarray(a, np.float64), 0.0, 10.0)
    b = np.clip(np.asarray(b, np.float64), 0.0, 10.0)
    return float(np.sqrt(((a - b) ** 2).mean()) / max(b.mean(), 1e-30))
