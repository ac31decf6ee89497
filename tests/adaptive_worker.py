"""One rank of the multi-GPU adaptive render (used by tests/test_adaptive_gpu.py; two of these share the test GPU and
exchange statistics over gloo). Writes the rank-0 result after the film reduction."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra  # noqa: E402
from raytracer_amd import _ffi, synth, api  # noqa: E402
from raytracer_amd.distributed import render_adaptive, reduce_film  # noqa: E402
from raytracer_amd.scene import render_params  # noqa: E402


def adaptive_frame(rank, world, W=96, H=54):
    ra.set_device(0)
    dev = torch.device("cuda:0")
    gs = ra.Scene(synth.build_scene(0.003))
    cam = synth.cerealbar_camera(W, H)
    rp = render_params(spp_min=8, spp_max=32, max_bounces=8, ci_halfwidth=0.02)
    n = W * H
    accum = torch.zeros((n, 4), dtype=torch.float32, device=dev)
    count = torch.zeros(n, dtype=torch.int32, device=dev)
    z = torch.zeros(n, dtype=torch.float32, device=dev)
    film = _ffi.Film(accum.data_ptr(), count.data_ptr(), z.data_ptr(), None, None)
    api.film_clear(film, W, H)
    s1 = torch.zeros((n, 4), dtype=torch.float32, device=dev)
    s2 = torch.zeros((n, 4), dtype=torch.float32, device=dev)
    pst = _ffi.PixelStats(s1.data_ptr(), s2.data_ptr())
    torch.cuda.synchronize()
    rounds = render_adaptive(gs, cam, rp, film, pst, s1, s2, rank=rank, world=world)
    torch.cuda.synchronize()
    if world > 1:
        a, c, zz = accum.cpu(), count.cpu(), z.cpu()     # gloo reduces host tensors
        reduce_film(a, c, zz, dst=0)
        accum.copy_(a); count.copy_(c); z.copy_(zz)
    res = api.film_resolve(film, cam) if rank == 0 else None
    return res, rounds


if __name__ == "__main__":
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res, rounds = adaptive_frame(rank, world)
    np.save(os.path.join(out, f"rounds_{rank}.npy"), np.array([[r["mine"], r["total"], r["samples"]] for r in rounds]))
    if rank == 0:
        np.savez(os.path.join(out, "result.npz"), linear=res["linear"], samples=res["samples"], z=res["z"])
    dist.barrier()
    dist.destroy_process_group()
