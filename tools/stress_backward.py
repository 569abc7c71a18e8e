"""Dev tool: run-to-run spread of the training step's gradients on identical inputs.  The plane and weight gradients are
accumulated with float atomics, so launches differ in summation order (~1e-7 of the element's magnitude); a sample whose
contribution is dropped or corrupted (see tests/test_gpu_determinism.py) shows up as a far larger per-element spread."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from triplaneturbo_amd import functional, ops

dev = torch.device("cuda", 0)
inp = bench.make_inputs(0, 1, dev, 1)
rc = ops.RenderConfig()
params = [inp["cache"]] + inp["sw"] + inp["fw"]
names = ["space_cache", "w1", "w2", "w3", "v1", "v2", "v3"]


def step():
    for t in params:
        t.grad = None
    out = functional.volume_render(inp["cache"], inp["sw"], inp["fw"], inp["ro"], inp["rd"], inp["ts"], inp["te"],
                                   inp["bg"], inp["cd"], inp["c2w"], rc, training=True)
    bench.loss_fn(out, inp["proj"]).backward()
    torch.cuda.synchronize()
    return [p.grad.clone() for p in params], {k: out[k].detach().clone() for k in ("comp_rgb", "opacity", "depth")}


ref, ref_out = step()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
worst = {k: 0.0 for k in names}
for it in range(n):
    g, o = step()
    for k in ref_out:
        if not torch.equal(ref_out[k], o[k]):
            print("forward output differs:", k, it, (ref_out[k] - o[k]).abs().max().item())
    for name, a, b in zip(names, ref, g):
        scale = a.abs().max().item()
        d = ((a - b).abs().max().item()) / scale
        worst[name] = max(worst[name], d)
for k, v in worst.items():
    print(f"{k:12s} max |g_run - g_0| / max|g_0| = {v:.3e}")
