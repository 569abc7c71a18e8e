"""Data-parallel glue: one process per GPU, prompts sharded across ranks (the reference's only parallelism is
Lightning DDP, configs/TriplaneTurbo_v1.yaml:255, launch.py:230-237).

d loss/d planes stays local to the rank (it flows on into that rank's SD-UNet/VAE backward); the only renderer-side
exchange is the sum of the six MLP weight gradients (16 640 fp32 = 66.6 KB).  That message is latency-bound, so it is
packed into ONE flat buffer and reduced with ONE RCCL all-reduce (xGMI is point-to-point: a single small collective,
not six per-tensor ones).  Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests.
"""
from __future__ import annotations

from typing import Iterable, List, Sequence

import torch


def shard_prompts(n_prompts: int, rank: int, world: int) -> range:
    """Contiguous split of the prompt batch (BASELINE config 4: 64 prompts -> 8 per GPU)."""
    base, rem = divmod(n_prompts, world)
    lo = rank * base + min(rank, rem)
    return range(lo, lo + base + (1 if rank < rem else 0))


def flatten_grads(params: Sequence[torch.Tensor]) -> torch.Tensor:
    return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])


def unflatten_into_grads(flat: torch.Tensor, params: Sequence[torch.Tensor]) -> None:
    ofs = 0
    for p in params:
        n = p.numel()
        g = flat[ofs:ofs + n].view_as(p)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        ofs += n


def allreduce_mlp_grads(params: Sequence[torch.Tensor], dist, average: bool = True) -> None:
    """Sum (DDP: average) the gradients of `params` over all ranks with one flat all-reduce."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return
    flat = flatten_grads(params)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if average:
        flat /= dist.get_world_size()
    unflatten_into_grads(flat, params)
