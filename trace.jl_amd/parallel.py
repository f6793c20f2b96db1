"""Multi-GPU sharding of one frame: rank r renders global sample indices [r*spp, (r+1)*spp) of every pixel
into a private film; the films are additive (film.jl:161-162, 190-191), so ONE collective ends the frame — a sum-reduce
to rank 0 (RCCL when the tensors live on GPUs, gloo in the CPU tests)."""
from __future__ import annotations


def shard_sample_offset(rank: int, spp_per_rank: int) -> int:
    return int(rank) * int(spp_per_rank)


def reduce_film(film, dst: int = 0):
    """Sum-reduce the (H, W, 4) film accumulators (xyz sums + filter_weight_sum) of all ranks onto `dst`, in place."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(film, dst=dst, op=dist.ReduceOp.SUM)
    return film
