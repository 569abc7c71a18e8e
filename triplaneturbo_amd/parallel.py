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


class FlatGradBucket:
    """The renderer-side parameters' gradients as views of ONE pre-flattened buffer (what DDP calls
    gradient_as_bucket_view): autograd accumulates straight into `p.grad` (= a view of `flat_grad`), so the
    per-backward exchange is a single in-place all-reduce of `flat_grad` -- no cat, no copy-back, no temporaries.

        bucket = FlatGradBucket(sdf_weights + feat_weights)    # once
        bucket.zero_()                                          # instead of p.grad = None
        loss.backward()
        bucket.all_reduce(dist)                                 # one RCCL call on the current stream
    """

    def __init__(self, params: Sequence[torch.Tensor]):
        self.params = list(params)
        if not self.params:
            raise ValueError("no parameters")
        p0 = self.params[0]
        n = sum(p.numel() for p in self.params)
        self.flat_grad = torch.zeros(n, dtype=p0.dtype, device=p0.device)
        self.bind()

    def bind(self) -> None:
        """(re)attach the views -- call again if something replaced p.grad (e.g. `p.grad = None`)"""
        ofs = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat_grad[ofs:ofs + n].view_as(p)
            ofs += n

    def zero_(self) -> None:
        self.flat_grad.zero_()
        if any(p.grad is None or p.grad.data_ptr() < self.flat_grad.data_ptr() or
               p.grad.data_ptr() >= self.flat_grad.data_ptr() + self.flat_grad.numel() * self.flat_grad.element_size()
               for p in self.params):
            self.bind()

    def all_reduce(self, dist, average: bool = True):
        """Sum (DDP: average) over ranks, in place.  Returns the async work handle when async_op is supported and
        wanted by the caller (`wait()` before the optimizer step); here the collective is enqueued on the current
        stream, behind the backward kernels that produced flat_grad, and the caller's next kernels queue behind it."""
        if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
            return None
        dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM)
        if average:
            self.flat_grad /= dist.get_world_size()
        return None
