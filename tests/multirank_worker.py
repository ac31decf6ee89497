"""One rank of the 2-rank GPU tests (tests/test_multirank_gpu.py): the ranks share the test GPU and talk over gloo, or
own a GPU each and talk over RCCL when the box has them (RTX_TEST_BACKEND=nccl). Exercises what bench.py and a
multi-GPU host use: render_accumulate + reduce_film over consecutive steps, and the photon-map all-gather."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra  # noqa: E402
from raytracer_amd import _ffi, synth, api  # noqa: E402
from raytracer_amd.distributed import reduce_film, all_gather_photons, photon_split  # noqa: E402
from raytracer_amd.scene import render_params  # noqa: E402

W, H, SPP, STEPS, NPHOT = 96, 54, 4, 3, 200000


def make_film(dev):
    n = W * H
    t = [torch.empty((n, 4), device=dev), torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, device=dev),
         torch.empty((n, 4), device=dev), torch.empty((n, 4), device=dev)]
    return t, _ffi.Film(*[x.data_ptr() for x in t])


def probes(ph):
    """Probe points on surfaces that hold photons, identical on every rank: the gathered records are the same set
    everywhere (in another order), so sort them and take every k-th position."""
    pts = []
    for which in (0, 1):
        pos = ph.download(which)["pos"]
        pos = pos[np.lexsort((pos[:, 2], pos[:, 1], pos[:, 0]))]
        pts.append(pos[:: max(1, len(pos) // 200)][:200] + np.float32(0.002))
    p = np.concatenate(pts, 0).astype(np.float32)
    return p, np.tile(np.array([[0, 1, 0]], np.float32), (len(p), 1))


def run(rank, world, local):
    ra.set_device(local)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(local)
    gs = ra.Scene(synth.build_scene(0.003))
    cam = synth.cerealbar_camera(W, H)
    out = {}
    # ---- consecutive frames: clear -> accumulate -> reduce, with different seeds; rank 0 keeps every reduced film ----
    tens, film = make_film(dev)
    for step in range(STEPS):
        rp = render_params(spp_min=SPP, max_bounces=8, rank=rank, world=world, seed=1000 + step)
        ra.film_clear(film, W, H)
        gs.render_accumulate(cam, rp, film)
        reduce_film(tens[0], tens[1], tens[2], dst=0, albedo=tens[3], normal=tens[4])
        if rank == 0:
            torch.cuda.synchronize()
            out[f"accum{step}"], out[f"count{step}"], out[f"z{step}"] = (t.cpu().numpy().copy() for t in tens[:3])
            out[f"albedo{step}"], out[f"normal{step}"] = tens[3].cpu().numpy().copy(), tens[4].cpu().numpy().copy()
    # ---- photon map: every rank traces its ids, then every rank holds the whole map ----
    ph = ra.Photons(gs, num_photons=NPHOT, **photon_split(rank, world))
    mine = ph.info()
    ng, nc = all_gather_photons(ph, device=dev)
    info = ph.info()
    assert (info["n_global"], info["n_caustic"]) == (ng, nc)
    p, n = probes(ph)
    out["probes"] = p
    out["mine"] = np.array([mine["n_global"], mine["n_caustic"]])
    out["total"] = np.array([ng, nc])
    out["est0"], out["est1"] = ph.estimate(0, p, n), ph.estimate(1, p, n)
    # the gathered caustic map drives the path loop's caustic term on every rank
    s, _ = gs.render_samples(synth.cerealbar_camera(32, 18), render_params(spp_min=2, max_bounces=4, use_photons=1), ph)
    out["caustic_samples"] = s
    return out


if __name__ == "__main__":
    rank, world, port, outdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    backend = os.environ.get("RTX_TEST_BACKEND", "gloo")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    local = rank if backend == "nccl" else 0
    force = bool(os.environ.get("RTX_AMD_DIST_FORCE"))   # a one-rank group whose collectives run all the same
    if world > 1 or force:
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    res = run(rank, world, local)
    np.savez(os.path.join(outdir, f"rank{rank}_of{world}{'_forced' if force else ''}.npz"), **res)
    if world > 1 or force:
        dist.barrier()
        dist.destroy_process_group()
