"""rtx_refine_list_noalloc (the entry point that allocates nothing; no counterpart in the reference) against
rtx_refine_list through the C ABI: identical ascending pixel lists on two calls in a row, the second on a smaller
frame (the static scratch keeps the first call's contents), and a clear refusal beyond the stated maximum."""
import ctypes as C

import numpy as np
import pytest

from raytracer_amd import _ffi, api

pytestmark = pytest.mark.gpu


def _random_stats(torch, W, H, spp_max, seed):
    """Per-pixel statistics as film_kernel leaves them: sum = {sum rgb, bits(n)}, sum2 = {sum of squares, running z}."""
    g = np.random.default_rng(seed)
    n = W * H
    cnt = g.integers(0, spp_max + 8, n).astype(np.int32)            # includes n = 0, 1 and n >= spp_max
    mean = g.uniform(0.0, 2.0, (n, 3)).astype(np.float32)
    sigma = (g.uniform(0.0, 0.25, (n, 1)) * (g.uniform(size=(n, 1)) < 0.7)).astype(np.float32)   # 30 % constant pixels
    fn = np.maximum(cnt, 1).astype(np.float32)[:, None]
    s1 = np.zeros((n, 4), np.float32)
    s2 = np.zeros((n, 4), np.float32)
    s1[:, :3] = mean * fn
    s2[:, :3] = (mean * mean + sigma * sigma) * fn
    s1[:, 3] = cnt.view(np.float32)
    s2[:, 3] = g.uniform(1.0, 9.0, n).astype(np.float32)
    dev = torch.device("cuda:0")
    return torch.from_numpy(s1).to(dev), torch.from_numpy(s2).to(dev)


def _both(torch, W, H, spp_max, ci, seed, capacity=None, stream=None):
    s1, s2 = _random_stats(torch, W, H, spp_max, seed)
    pst = _ffi.PixelStats(s1.data_ptr(), s2.data_ptr())
    torch.cuda.synchronize()            # torch has filled the inputs (its own runtime: a host wait is the only ordering there is)
    want = api.refine_list(pst, W, H, ci, spp_max)
    cap = W * H if capacity is None else capacity
    out = torch.full((max(cap, 1),), -7, dtype=torch.int32, device=s1.device)
    cnt = torch.zeros(1, dtype=torch.int32, device=s1.device)
    torch.cuda.synchronize()
    api.refine_list_noalloc(pst, W, H, ci, spp_max, out.data_ptr(), cap, cnt.data_ptr(), stream)
    lib = _ffi.load_library()
    # The entry point synchronises nothing: the caller does - with the LIBRARY's runtime, on which the kernels were enqueued
    # (torch.cuda.synchronize() waits for torch's own runtime only).
    assert (lib.rtx_stream_synchronize(stream) if stream else lib.rtx_device_synchronize()) == 0, lib.rtx_last_error()
    return want, out.cpu().numpy(), int(cnt.item())


def test_noalloc_matches_refine_list_twice_in_a_row(rtx):
    import torch
    rtx.set_device(0)
    # first call: a 1080p frame (2025 compaction blocks); second call: a smaller frame that is no multiple of the block
    want, got, cnt = _both(torch, 1920, 1080, 64, 0.02, seed=1)
    assert 0 < len(want) < 1920 * 1080
    assert cnt == len(want) and np.array_equal(got[:cnt], want) and (got[cnt:] == -7).all()
    want2, got2, cnt2 = _both(torch, 333, 97, 32, 0.01, seed=2)
    assert 0 < len(want2) < 333 * 97
    assert cnt2 == len(want2) and np.array_equal(got2[:cnt2], want2) and (got2[cnt2:] == -7).all()
    # a capacity below the count: the count is still the full one, the list is its ascending prefix
    want3, got3, cnt3 = _both(torch, 333, 97, 32, 0.01, seed=2, capacity=100)
    assert cnt3 == len(want3) and np.array_equal(got3[:100], want3[:100])
    # nothing to refine
    want4, got4, cnt4 = _both(torch, 64, 64, 2, 0.5, seed=3)
    assert len(want4) == 0 and cnt4 == 0 and (got4 == -7).all()
    # on a stream of the library's runtime instead of the null stream, twice (the shared scratch is reused in stream order)
    lib = _ffi.load_library()
    st = C.c_void_p()
    assert lib.rtx_stream_create(C.byref(st)) == 0, lib.rtx_last_error()
    try:
        for seed, (W, H) in ((4, (640, 360)), (5, (257, 129))):
            w5, g5, c5 = _both(torch, W, H, 48, 0.015, seed=seed, stream=st)
            assert 0 < len(w5) and c5 == len(w5) and np.array_equal(g5[:c5], w5) and (g5[c5:] == -7).all()
    finally:
        assert lib.rtx_stream_destroy(st) == 0


def test_noalloc_rejects_oversize_and_bad_arguments(rtx):
    import torch
    lib = _ffi.load_library()
    s = torch.zeros((16, 4), dtype=torch.float32, device="cuda:0")
    pst = _ffi.PixelStats(s.data_ptr(), s.data_ptr())
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda:0")
    # 16384 x 8192 = 2^27 pixels > RTX_REFINE_NOALLOC_MAX_PIXELS: refused before anything is launched
    rc = lib.rtx_refine_list_noalloc(C.byref(pst), 16384, 8192, 0.005, 1024, None, 0, cnt.data_ptr(), None)
    assert rc == -4 and b"RTX_REFINE_NOALLOC_MAX_PIXELS" in lib.rtx_last_error()
    assert lib.rtx_refine_list_noalloc(C.byref(pst), 4, 4, 0.005, 1024, None, 0, None, None) == -1
    assert lib.rtx_refine_list_noalloc(C.byref(pst), 4, 4, 0.005, 1024, None, 8, cnt.data_ptr(), None) == -1
    assert lib.rtx_refine_list_noalloc(None, 4, 4, 0.005, 1024, None, 0, cnt.data_ptr(), None) == -1
