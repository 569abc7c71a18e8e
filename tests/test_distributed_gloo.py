"""world_size-2 CPU test (gloo) of the data-parallel glue: prompts sharded per rank, ONE flat all-reduce of the six
MLP gradients averaged DDP-style (the reference relies on Lightning DDP, configs/TriplaneTurbo_v1.yaml:255)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from triplaneturbo_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)  # identical parameters on every rank (as DDP guarantees)
    shapes = [(64, 32), (64, 64), (1, 64), (64, 96), (64, 64), (3, 64)]
    params = [torch.randn(s, requires_grad=True) for s in shapes]
    g = torch.Generator().manual_seed(100 + rank)  # different prompts per rank => different local grads
    local = [torch.randn(s, generator=g) for s in shapes]
    for p, l in zip(params, local):
        p.grad = l.clone()
    params[2].grad = None  # a parameter without a local gradient must still take part
    parallel.allreduce_mlp_grads(params, dist)
    calls = []  # exactly one collective: patch and repeat
    orig = dist.all_reduce
    dist.all_reduce = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    parallel.allreduce_mlp_grads(params, dist, average=False)
    dist.all_reduce = orig
    # flat-bucket path: gradients accumulate by autograd into views of one buffer; one in-place collective, no cat
    params2 = [torch.randn(s, requires_grad=True) for s in shapes]
    bucket = parallel.FlatGradBucket(params2)
    bucket.zero_()
    loss = sum((p * l).sum() for p, l in zip(params2, local))  # d loss / d p = local
    loss.backward()
    assert all(p.grad.data_ptr() >= bucket.flat_grad.data_ptr() for p in params2)  # still the views
    calls2 = []
    dist.all_reduce = lambda *a, **k: (calls2.append(a[0].data_ptr()), orig(*a, **k))[1]
    bucket.all_reduce(dist)
    dist.all_reduce = orig
    flat_ok = calls2 == [bucket.flat_grad.data_ptr()]
    params2[1].grad = None  # somebody dropped a view: zero_() re-binds it
    bucket.zero_()
    rebound = params2[1].grad is not None and float(bucket.flat_grad.abs().sum()) == 0.0
    # plain lists: tensors in an mp.Queue are shared-memory handles that die with the worker
    loss = sum((p * l).sum() for p, l in zip(params2, local))
    loss.backward()
    bucket.all_reduce(dist)
    q.put((rank, [p.grad.tolist() for p in params], [l.tolist() for l in local], len(calls),
           list(parallel.shard_prompts(7, rank, world)), [p.grad.tolist() for p in params2], flat_ok and rebound))
    dist.barrier()
    dist.destroy_process_group()


def test_allreduce_mlp_grads_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, g0, l0, n0, s0, f0, ok0), (_, g1, l1, n1, s1, f1, ok1) = res
    g0, l0, g1, l1, f0, f1 = [[torch.tensor(t) for t in x] for x in (g0, l0, g1, l1, f0, f1)]
    assert ok0 and ok1  # one collective, on the flat buffer itself; dropped views are re-bound
    for i in range(6):
        torch.testing.assert_close(f0[i], (l0[i] + l1[i]) / 2, rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(f0[i], f1[i], rtol=0, atol=0)
    assert n0 == n1 == 1
    assert s0 + s1 == list(range(7))
    for i in range(6):
        a = torch.zeros_like(l0[i]) if i == 2 else l0[i]
        b = torch.zeros_like(l1[i]) if i == 2 else l1[i]
        mean = (a + b) / 2
        # first call averaged; second call (average=False) summed the already-identical averages
        torch.testing.assert_close(g0[i], 2 * mean, rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(g0[i], g1[i], rtol=0, atol=0)
