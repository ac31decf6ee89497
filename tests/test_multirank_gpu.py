"""Two ranks of the HIP path against one rank: (1) render_accumulate + reduce_film over consecutive
frames equals the single-rank film (counts and z exact, colour up to float summation order, guides exact) - this is the
loop bench.py times; (2) the photon map all-gathered from two ranks that traced half of the ids each equals the
single-rank map: stored counts exact, irradiance estimates within 1e-3, identical caustic term in the path loop.
The ranks share the test GPU and use gloo (the nccl branch needs one GPU per rank: RTX_TEST_BACKEND=nccl)."""
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _spawn(world, tmp):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = str(s.getsockname()[1])
    s.close()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "multirank_worker.py"), str(r), str(world), port, tmp],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=280)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-3000:] for o in outs)
    return [dict(np.load(os.path.join(tmp, f"rank{r}_of{world}.npz"))) for r in range(world)]


@pytest.fixture(scope="module")
def runs(rtx):
    with tempfile.TemporaryDirectory() as tmp:
        one = _spawn(1, tmp)[0]
        two = _spawn(2, tmp)
    return one, two


def test_consecutive_frames_reduce_to_the_single_rank_film(runs):
    one, (r0, r1) = runs
    for step in range(3):
        assert np.array_equal(r0[f"count{step}"], one[f"count{step}"]) and (one[f"count{step}"] == 4).all()
        assert np.array_equal(r0[f"z{step}"], one[f"z{step}"])
        assert np.allclose(r0[f"accum{step}"], one[f"accum{step}"], rtol=1e-4, atol=1e-5)
        assert np.array_equal(r0[f"albedo{step}"], one[f"albedo{step}"])
        assert np.array_equal(r0[f"normal{step}"], one[f"normal{step}"])
    # the steps differ (different seeds): a stale or half-cleared film would show up as a step repeating
    assert not np.allclose(one["accum0"], one["accum1"])


def test_all_gathered_photon_map_equals_single_rank_map(runs):
    one, (r0, r1) = runs
    assert (r0["mine"] + r1["mine"] == one["total"]).all()             # ids are split, nothing lost or doubled
    assert (r0["total"] == one["total"]).all() and (r1["total"] == one["total"]).all()
    assert (r0["mine"] > 0).all() and (r1["mine"] > 0).all()
    assert np.array_equal(r0["probes"], one["probes"]) and np.array_equal(r1["probes"], one["probes"])
    for k in ("est0", "est1"):
        ref = one[k]
        assert np.abs(ref).max() > 0
        for r in (r0, r1):
            assert np.allclose(r[k], ref, rtol=1e-3, atol=1e-6 * np.abs(ref).max())
    for r in (r0, r1):
        d = np.abs(r["caustic_samples"] - one["caustic_samples"]).max(-1)
        assert (d <= 1e-4 * (1 + np.abs(one["caustic_samples"]).max(-1))).mean() >= 0.999
