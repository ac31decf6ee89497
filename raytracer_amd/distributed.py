"""Multi-GPU plumbing: one process per GPU, tiles dealt round-robin, one film reduction per frame.

The reference merges per-thread full-frame buffers at frame end (RayCaster.cpp:1182-1212: sum of colour, weight and
sample counts, min of z). Ranks do the same across GPUs with torch.distributed (backend "nccl" = RCCL over xGMI on
ROCm; "gloo" in the CPU tests). Samples splat into neighbouring pixels (tent radius 1, RayCaster.cpp:810-829), so the
exchange must be a sum-reduce of full-frame films, not a gather of disjoint tiles.
"""
import torch.distributed as dist


def reduce_film(accum, count, z, dst=0, group=None):
    """In-place reduction of the film tensors to rank `dst`: accum [n,4] and count [n] by SUM, z [n] by MIN."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    dist.reduce(accum, dst, op=dist.ReduceOp.SUM, group=group)
    dist.reduce(count, dst, op=dist.ReduceOp.SUM, group=group)
    dist.reduce(z, dst, op=dist.ReduceOp.MIN, group=group)


def photon_split(rank, world):
    """Photon ids of a rank: rank, rank + world, ... (PhotonMapBuilder.cpp:110 hands ids out one by one)."""
    return dict(first_photon=rank, photon_stride=world)
