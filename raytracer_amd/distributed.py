"""Multi-GPU plumbing: one process per GPU, tiles dealt round-robin, one film reduction per frame.

The reference merges per-thread full-frame buffers at frame end (RayCaster.cpp:1182-1212: sum of colour, weight and
sample counts, min of z). Ranks do the same across GPUs with torch.distributed (backend "nccl" = RCCL over xGMI on
ROCm; "gloo" in the CPU tests). Samples splat into neighbouring pixels (tent radius 1, RayCaster.cpp:810-829), so the
exchange must be a sum-reduce of full-frame films, not a gather of disjoint tiles.
"""
import os

import numpy as np
import torch
import torch.distributed as dist


def _alone(group):
    """No exchange needed: no process group, or a group of one. RTX_AMD_DIST_FORCE=1 runs the collectives of a
    one-rank group anyway (the RCCL code path on a box with a single GPU: tests/test_multirank_gpu.py)."""
    if not dist.is_initialized():
        return True
    return dist.get_world_size(group) == 1 and not os.environ.get("RTX_AMD_DIST_FORCE")


def _staged(t, group):
    """gloo moves host memory: CUDA tensors go through the host (CPU tests, rehearsals on a box with fewer GPUs)."""
    return dist.get_backend(group) == "gloo" and t.is_cuda


def reduce_film(accum, count, z, dst=0, group=None, albedo=None, normal=None):
    """In-place reduction of the film tensors to rank `dst`: accum [n,4] and count [n] by SUM, z [n] by MIN; the
    first-hit guides albedo / normal [n,4], when given, by SUM (a pixel's guides are written only by the rank that owns
    its bucket, zero elsewhere: include/rtx.h rtx_film).

    Returns with the reduction COMPLETE on the host side: over RCCL, dist.reduce only orders the collective on torch's
    current stream, and the next consumer of these buffers is librtx_amd (rtx_film_clear / rtx_film_resolve of the next
    step), which runs on the null stream of its own HIP runtime and has no stream ordering with torch. Without the
    synchronisation below the next step's clear could zero the film while the reduce still reads or writes it."""
    if _alone(group):
        return
    jobs = [(accum, dist.ReduceOp.SUM), (count, dist.ReduceOp.SUM), (z, dist.ReduceOp.MIN)]
    jobs += [(t, dist.ReduceOp.SUM) for t in (albedo, normal) if t is not None]
    any_cuda = False
    for t, op in jobs:
        if _staged(t, group):
            h = t.cpu()
            dist.reduce(h, dst, op=op, group=group)
            if dist.get_rank(group) == dst:
                t.copy_(h)
        else:
            dist.reduce(t, dst, op=op, group=group)
        any_cuda = any_cuda or t.is_cuda
    if any_cuda:
        torch.cuda.synchronize()


def photon_split(rank, world):
    """Photon ids of a rank: rank, rank + world, ... (PhotonMapBuilder.cpp:110 hands ids out one by one)."""
    return dict(first_photon=rank, photon_stride=world)


def all_gather_records(local, group=None):
    """All-gather of per-rank record arrays of different lengths. local: [n_rank, k] tensor (any n_rank >= 0, same k
    and dtype on every rank). Returns ([sum n, k] tensor with the ranks' records in rank order, list of per-rank counts).
    Sizes are exchanged first, then one padded all_gather moves the records (RCCL all-gathers equal-sized buffers)."""
    world = dist.get_world_size(group)
    staged = _staged(local, group)
    sdev = torch.device("cpu") if (staged or not local.is_cuda) else local.device
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=sdev)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    counts = [int(x.item()) for x in sizes]
    nmax = max(counts)
    if nmax == 0:
        return local[:0].clone(), counts
    send = torch.zeros((nmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=sdev)
    send[:local.shape[0]] = local.to(sdev)
    recv = torch.empty((world, nmax) + tuple(local.shape[1:]), dtype=local.dtype, device=sdev)
    if sdev.type == "cuda":
        dist.all_gather_into_tensor(recv.view((world * nmax,) + tuple(local.shape[1:])), send, group=group)
    else:
        dist.all_gather(list(recv.unbind(0)), send, group=group)
    out = torch.cat([recv[r, :counts[r]] for r in range(world)], dim=0).to(local.device)
    return out.contiguous(), counts


def all_gather_photons(photons, group=None, device=None):
    """Every rank has traced its share of the photon ids (Photons(..., **photon_split(rank, world))); afterwards every
    rank holds the whole map, as in the reference, where all workers feed one map that is scaled once
    (PhotonMapBuilder.cpp:102-146, :38-50): the raw 48-B records of both sets are all-gathered (RCCL over xGMI; gloo
    stages through the host) and rtx_photon_set rebuilds the hash grid and the power scaling (global map: 1/emitted,
    caustic map: 1/stored over ALL ranks) on every rank. Returns (n_global, n_caustic) of the gathered map."""
    if _alone(group):
        info = photons.info()
        return info["n_global"], info["n_caustic"]
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    totals = []
    for which, key in ((0, "n_global"), (1, "n_caustic")):
        n_local = photons.info()[key]
        local = torch.empty((n_local, 12), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()                      # the buffer exists before the library's own runtime writes it
        if n_local:
            photons.copy_raw(which, local.data_ptr(), n_local)
        allrec, counts = all_gather_records(local, group)
        torch.cuda.synchronize()                      # gathered records are in place before librtx_amd reads them
        photons.set_raw(which, allrec.data_ptr() if allrec.shape[0] else None, int(allrec.shape[0]), photons.num_photons)
        totals.append(int(allrec.shape[0]))
    return tuple(totals)


# ---- adaptive sampling across GPUs -----------------------------------------------------
# Round 1 (spp_min samples everywhere) uses the static round-robin bucket split. After every round the ranks share
# the per-pixel statistics of the pixels they just sampled, run the same confidence-interval test
# (RayCaster.cpp:848-868) and deal the buckets that still hold unconverged pixels out again, heaviest first, to the
# least loaded rank (greedy LPT). Migrating a bucket costs nothing beyond that exchange: its accumulated colour
# stays in the film of whichever rank produced it (films are summed at frame end), only the statistics move.


def bucket_of_pixels(pixels, width, height, tile=16):
    """Bucket index (row-major over ceil(W/tile) x ceil(H/tile), RayCaster.cpp:1152-1158) of pixel indices y*W+x."""
    pixels = np.asarray(pixels, dtype=np.int64)
    nbx = (width + tile - 1) // tile
    return ((pixels // width) // tile) * nbx + (pixels % width) // tile


def lpt_assign(weights, world):
    """Greedy longest-processing-time assignment. weights: array of per-bucket work (0 = nothing to do).
    Returns owner[bucket] in [0, world) for buckets with work and -1 otherwise. Deterministic: buckets by decreasing
    weight then increasing index, ties between ranks to the lowest rank."""
    weights = np.asarray(weights, dtype=np.int64)
    owner = np.full(weights.shape[0], -1, dtype=np.int32)
    load = np.zeros(world, dtype=np.int64)
    busy = np.nonzero(weights > 0)[0]
    order = busy[np.lexsort((busy, -weights[busy]))]
    for b in order:
        r = int(np.argmin(load))
        owner[b] = r
        load[r] += weights[b]
    return owner


def _all_reduce_exact(t, group=None):
    """SUM all-reduce of bit patterns (int32 view): exact when at most one rank holds a non-zero value per element."""
    v = t.view(torch.int32)
    if dist.get_backend(group) == "gloo" and v.is_cuda:
        h = v.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        v.copy_(h)
    else:
        dist.all_reduce(v, op=dist.ReduceOp.SUM, group=group)


def render_adaptive(scene, cam, params, film, pstats, stat_sum, stat_sum2, rank=0, world=1, group=None, photons=None,
                    tile=16):
    """Adaptive CastRays (spp_min ... spp_max, RayCaster.cpp:770-980) on `world` GPUs with per-round LPT re-balancing.

    scene: raytracer_amd.Scene; film: _ffi.Film over this rank's device buffers (cleared by the caller);
    pstats: _ffi.PixelStats over stat_sum / stat_sum2 (zero-filled float32 [W*H, 4] device tensors).
    Returns a list with one dict per round: pixels sampled by this rank, by all ranks, samples per pixel."""
    from . import api
    W, H = cam.width, cam.height
    spp_min, spp_max = params.spp_min, params.spp_max
    multi = world > 1 and dist.is_initialized()
    mine = api.pixel_list(W, H, tile, rank, world)
    everyone = None            # pixels sampled in the round just finished, over all ranks (None = the whole frame)
    base, rounds = 0, []
    while True:
        n = min(spp_min, spp_max - base)
        if n <= 0:
            break
        scene.render_round(cam, params, film, pstats, mine, base, n, reset_z=(len(rounds) == 1), photons=photons)
        rounds.append(dict(mine=int(mine.shape[0]), total=int(W * H if everyone is None else everyone.shape[0]), samples=n))
        base += n
        if spp_max <= spp_min or base >= spp_max:
            break
        if multi:
            # statistics of the pixels this rank just sampled -> every rank
            for t in (stat_sum, stat_sum2):
                contrib = torch.zeros_like(t)
                if mine.shape[0]:
                    idx = torch.from_numpy(mine.astype(np.int64)).to(t.device)
                    contrib[idx] = t[idx]
                _all_reduce_exact(contrib, group)
                if everyone is None:
                    t.copy_(contrib)
                else:
                    idx = torch.from_numpy(everyone.astype(np.int64)).to(t.device)
                    t[idx] = contrib[idx]
            torch.cuda.synchronize()
        refine = api.refine_list(pstats, W, H, params.ci_halfwidth, spp_max)
        if refine.shape[0] == 0:
            break
        buckets = bucket_of_pixels(refine, W, H, tile)
        nb = ((W + tile - 1) // tile) * ((H + tile - 1) // tile)
        owner = lpt_assign(np.bincount(buckets, minlength=nb), world)
        mine = refine[owner[buckets] == rank]
        everyone = refine
    return rounds
