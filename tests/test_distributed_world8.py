"""World-size-8 plumbing that needs no GPU: the prompt shard of BASELINE configs[3] (64 prompts over 8 ranks) and the flat
MLP-gradient bucket, over gloo on CPU tensors -- sum of the ranks' shard gradients / world == the unsharded batch's mean
gradient, every prompt rendered exactly once."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from triplaneturbo_amd.parallel import FlatGradBucket, shard_prompts

WORLD, N_PROMPTS = 8, 64


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _params():
    g = torch.Generator().manual_seed(0)
    return [torch.randn(s, generator=g, dtype=torch.float64).requires_grad_(True) for s in ((64, 32), (64, 64), (1, 64), (64, 96), (64, 64), (3, 64))]


def _prompt_loss(params, p):
    """a stand-in for one prompt's render loss: any smooth function of the six matrices that depends on the prompt"""
    g = torch.Generator().manual_seed(100 + p)
    x = torch.randn(5, 32, generator=g, dtype=torch.float64)
    e = torch.randn(5, 96, generator=g, dtype=torch.float64)
    h = torch.relu(torch.relu(x @ params[0].T) @ params[1].T) @ params[2].T
    k = torch.relu(torch.relu(e @ params[3].T) @ params[4].T) @ params[5].T
    return (h ** 2).sum() + k.sum()


def _rank(rank, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        params = _params()
        bucket = FlatGradBucket(params)
        mine = list(shard_prompts(N_PROMPTS, rank, WORLD))
        bucket.zero_()
        loss = sum(_prompt_loss(params, p) for p in mine) / len(mine)
        loss.backward()
        bucket.all_reduce(dist)  # one flat in-place all-reduce, averaged over the ranks
        seen = [None] * WORLD
        dist.all_gather_object(seen, mine)
        torch.save({"grad": bucket.flat_grad.clone(), "seen": seen, "world": dist.get_world_size()},
                   os.path.join(out_dir, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_64_prompts_over_8_ranks_equal_the_unsharded_batch(tmp_path):
    mp.spawn(_rank, args=(_free_port(), str(tmp_path)), nprocs=WORLD, join=True)
    res = [torch.load(os.path.join(tmp_path, f"r{r}.pt")) for r in range(WORLD)]
    assert all(r["world"] == WORLD for r in res)
    seen = sorted(p for shard in res[0]["seen"] for p in shard)
    assert seen == list(range(N_PROMPTS))  # every prompt exactly once
    assert all(len(s) == N_PROMPTS // WORLD for s in res[0]["seen"])
    params = _params()
    loss = sum(_prompt_loss(params, p) for p in range(N_PROMPTS)) / N_PROMPTS
    ref = torch.cat([g.reshape(-1) for g in torch.autograd.grad(loss, params)])
    for r in res:
        torch.testing.assert_close(r["grad"], ref, rtol=1e-11, atol=1e-12)  # (fp64: only the summation order differs)  # identical on every rank, = the unsharded mean


@pytest.mark.parametrize("n,world", [(64, 8), (8, 8), (10, 4), (7, 8)])
def test_shard_prompts_partitions(n, world):
    shards = [list(shard_prompts(n, r, world)) for r in range(world)]
    assert sorted(p for s in shards for p in s) == list(range(n))
    assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1
    assert all(s == list(range(s[0], s[0] + len(s))) for s in shards if s)  # contiguous
