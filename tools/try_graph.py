"""Dev tool: capture one bench step (fwd + loss + bwd) in a hipGraph via torch.cuda.graph and time replays vs eager."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from triplaneturbo_amd import functional, ops  # noqa: E402

dev = torch.device("cuda", 0)
inp = bench.make_inputs(0, 1, dev, 1)
rc = ops.RenderConfig()
params = [inp["cache"]] + inp["sw"] + inp["fw"]
for p in params:
    p.grad = torch.zeros_like(p)  # static gradient buffers (graph replays write into them)


def step():
    out = functional.volume_render(inp["cache"], inp["sw"], inp["fw"], inp["ro"], inp["rd"], inp["ts"], inp["te"],
                                   inp["bg"], inp["cd"], inp["c2w"], rc, training=True)
    loss = bench.loss_fn(out, inp["proj"])
    grads = torch.autograd.grad(loss, params)
    for p, g in zip(params, grads):
        p.grad.copy_(g)
    return loss


def timeit(fn, n=20):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print("eager ms/step", round(timeit(step), 3), flush=True)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    static_loss = step()
torch.cuda.synchronize()
ref = [p.grad.clone() for p in params]
g.replay()
torch.cuda.synchronize()
print("replay loss", float(static_loss), "grad equal-ish:",
      [round(((a - b.grad).norm() / a.norm()).item(), 8) for a, b in zip(ref, params)], flush=True)
print("graph ms/step", round(timeit(g.replay), 3), flush=True)
