"""world_size-2 test of the multi-rank path on CPU (gloo): each rank renders the buckets it owns into a full-frame
film, the films are reduced with raytracer_amd.distributed.reduce_film (the code bench.py uses over RCCL), and the
result equals the single-rank render. The oracle stands in for the per-rank renderer; the reduction protocol and the
bucket ownership rule are the parts under test."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracer_amd import synth
from raytracer_amd.distributed import reduce_film, photon_split, all_gather_records
from raytracer_amd.scene import render_params
from tests.oracle_ffi import OracleScene, OraclePhotons


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sd = synth.build_scene(0.002)
    cam = synth.cerealbar_camera(48, 27)
    osc = OracleScene(sd)
    r = osc.render(cam, render_params(spp_min=4, max_bounces=8, rank=rank, world=world))
    n = 48 * 27
    accum = torch.from_numpy(r["accum"].reshape(n, 4).copy())
    count = torch.from_numpy(r["samples"].reshape(n).copy())
    z = torch.from_numpy(r["z"].reshape(n).copy())
    own = int(count.sum())
    reduce_film(accum, count, z, dst=0)
    # photon ids are split by stride; counts add up
    ph = OraclePhotons(osc, 20000, finalize=False, **photon_split(rank, world))
    cnt = torch.tensor([ph.info()["n_global"], ph.info()["n_caustic"], own], dtype=torch.int64)
    dist.all_reduce(cnt)
    if rank == 0:
        np.savez(out_path, accum=accum.numpy(), count=count.numpy(), z=z.numpy(), cnt=cnt.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_film_reduce_equals_single_rank(tmp_path):
    out = str(tmp_path / "r.npz")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    sd = synth.build_scene(0.002)
    cam = synth.cerealbar_camera(48, 27)
    osc = OracleScene(sd)
    ref = osc.render(cam, render_params(spp_min=4, max_bounces=8))
    n = 48 * 27
    assert np.array_equal(got["count"], ref["samples"].reshape(n))          # every pixel sampled by exactly one rank
    assert got["cnt"][2] == n * 4
    assert np.array_equal(got["z"], ref["z"].reshape(n))
    assert np.allclose(got["accum"], ref["accum"].reshape(n, 4), rtol=1e-5, atol=1e-6)   # float sum order only
    ph = OraclePhotons(osc, 20000, finalize=False)
    assert got["cnt"][0] == ph.info()["n_global"] and got["cnt"][1] == ph.info()["n_caustic"]


def _gather_worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(100 + rank)
    res = {}
    # ragged photon record arrays (48-B raw records = 12 floats): different lengths per rank, an empty rank, all empty
    for case, lens in (("ragged", (5, 3)), ("empty_rank", (0, 7)), ("all_empty", (0, 0))):
        local = torch.rand((lens[rank], 12), generator=g)
        allrec, counts = all_gather_records(local)
        res[case] = allrec.numpy()
        res[case + "_counts"] = np.array(counts)
        res[case + "_local"] = local.numpy()
    # film guides ride along with the reduce: each pixel written by exactly one rank
    n = 64
    own = (torch.arange(n) % world) == rank
    accum = torch.rand((n, 4), generator=g)
    count = own.to(torch.int32) * 4
    z = torch.where(own, torch.rand(n, generator=g), torch.full((n,), 1.0e30))
    albedo = torch.where(own[:, None], torch.rand((n, 4), generator=g), torch.zeros((n, 4)))
    res["albedo_local"], res["accum_local"], res["z_local"] = albedo.numpy().copy(), accum.numpy().copy(), z.numpy().copy()
    reduce_film(accum, count, z, dst=0, albedo=albedo)
    res["albedo"], res["accum"], res["z"], res["count"] = albedo.numpy(), accum.numpy(), z.numpy(), count.numpy()
    np.savez(out_path % rank, **res)
    dist.barrier()
    dist.destroy_process_group()


def test_record_all_gather_and_guide_reduce(tmp_path):
    """distributed.all_gather_records (the transport of all_gather_photons) and reduce_film's guide buffers, world 2."""
    out = str(tmp_path / "g%d.npz")
    mp.spawn(_gather_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r = [np.load(out % k) for k in range(2)]
    for case in ("ragged", "empty_rank", "all_empty"):
        want = np.concatenate([r[0][case + "_local"], r[1][case + "_local"]], 0)
        for k in range(2):                                        # every rank holds every record, in rank order
            assert np.array_equal(r[k][case], want)
            assert list(r[k][case + "_counts"]) == [len(r[0][case + "_local"]), len(r[1][case + "_local"])]
    assert np.array_equal(r[0]["albedo"], r[0]["albedo_local"] + r[1]["albedo_local"])
    assert np.allclose(r[0]["accum"], r[0]["accum_local"] + r[1]["accum_local"])
    assert np.array_equal(r[0]["z"], np.minimum(r[0]["z_local"], r[1]["z_local"]))
    assert (r[0]["count"] == 4).all()


def test_lpt_assignment_is_balanced_and_deterministic():
    """raytracer_amd.distributed.lpt_assign / bucket_of_pixels: the host logic of the per-round re-balancing (8e)."""
    from raytracer_amd.distributed import lpt_assign, bucket_of_pixels
    from raytracer_amd import api
    rng = np.random.default_rng(3)
    w = rng.integers(0, 257, 8160)
    w[rng.random(8160) < 0.6] = 0                      # most buckets converge after round 1
    for world in (1, 2, 4, 8):
        owner = lpt_assign(w, world)
        assert ((owner >= 0) == (w > 0)).all() and owner.max() < world
        load = np.bincount(owner[owner >= 0], weights=w[owner >= 0], minlength=world)
        assert load.max() - load.min() <= w.max()      # greedy LPT bound
        assert np.array_equal(owner, lpt_assign(w.copy(), world))
        # the C++ group path (rtx_group_render between adaptive rounds) deals by the same rule
        from raytracer_amd import _ffi
        w32 = np.ascontiguousarray(w, np.int32)
        own_c = np.full(w32.shape[0], -7, np.int32)
        assert _ffi.load_library().rtx_debug_lpt_assign(w32.ctypes.data, w32.shape[0], world, own_c.ctypes.data) == 0
        assert np.array_equal(own_c, owner)
    # a clustered load that the static b % world split handles badly
    w2 = np.zeros(8160, np.int64)
    w2[0:400:8] = 256
    static = np.bincount(np.arange(8160)[w2 > 0] % 8, weights=w2[w2 > 0], minlength=8)
    lpt = np.bincount(lpt_assign(w2, 8)[w2 > 0], weights=w2[w2 > 0], minlength=8)
    assert static.max() == w2.sum() and lpt.max() <= w2.sum() / 8 + 256
    # bucket ids agree with the library's bucket order
    W, H = 100, 70
    for rank in range(3):
        px = api.pixel_list(W, H, 16, rank, 3)
        assert (bucket_of_pixels(px, W, H, 16) % 3 == rank).all()
    allpx = np.sort(np.concatenate([api.pixel_list(W, H, 16, r, 3) for r in range(3)]))
    assert np.array_equal(allpx, np.arange(W * H))
