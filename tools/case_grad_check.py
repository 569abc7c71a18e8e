"""Dev tool: gradients of one test_backward_matches_oracle-style case in the three precision modes against the fp32 / fp64
oracle, optionally on an experiment build of the library.  usage: python tools/case_grad_check.py [--variant NAME] P R n_view Hh Ww S seed"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
args = sys.argv[1:]
from triplaneturbo_amd import _lib  # noqa: E402
if args and args[0] == "--variant":
    _lib.use_variant(args[1])
    args = args[2:]
from oracle import cpu_ref as O  # noqa: E402
from test_gpu_backward import KEYS, _hip_grads, _oracle_grads  # noqa: E402
from parity import rel  # noqa: E402
from triplaneturbo_amd import functional, ops  # noqa: E402

P, R, n_view, Hh, Ww, S, seed = [int(a) for a in args]
g = torch.Generator().manual_seed(seed)
cache = torch.randn(P, 6, 32, R, R, generator=g) * 0.5
sw = O.init_mlp_weights([32, 64, 64, 1], g)
fw = O.init_mlp_weights([96, 64, 64, 3], g)
ro, rd, c2w, cd = O.make_cameras(P * n_view, Hh, Ww)
ts, te = O.uniform_intervals(P * n_view * Hh * Ww, S, 0.3, 3.2)
bg = torch.ones(3)
proj = {n: torch.randn(P * n_view, Hh, Ww, c, generator=g) for n, c in KEYS}
rck = dict(inv_std=100.0, rgb_grad_shrink=0.7, cos_anneal_ratio=1.0)
a = (cache, sw, fw, ro, rd, ts, te, bg, cd, c2w, proj)
_, _, g32 = _oracle_grads(torch.float32, *a, rck)
_, _, g64 = _oracle_grads(torch.float64, *a, rck)
alts = [_oracle_grads(torch.float32, *a, rck, alt_order=lv)[2] for lv in (1, 2, 3)]
print("fp32 vs fp64 %.2e | fp32 alternatives vs fp32: %s" % (rel(g32[0], g64[0]), " ".join("%.2e" % rel(a_[0], g32[0]) for a_ in alts)))
for mode in ("split3", "f32", "split2"):
    _, _, gh = _hip_grads((ops, functional), *a, dict(rck, precision=mode))
    print(mode, " ".join("%.2e" % rel(x, y) for x, y in zip(gh, g32)))

# per output key: gradient of <out[k], proj[k]> w.r.t. the planes, HIP (split3) against the fp32 oracle; and forward outputs
dev = "cuda"
c = cache.to(dev).requires_grad_(True)
sws = [w.to(dev).requires_grad_(True) for w in sw]
fws = [w.to(dev).requires_grad_(True) for w in fw]
out = functional.volume_render(c, sws, fws, ro.to(dev), rd.to(dev), ts.to(dev), te.to(dev), bg.to(dev), cd.to(dev), c2w.to(dev),
                               ops.RenderConfig(**dict(rck, precision="split3")), training=True)
c32 = cache.clone().requires_grad_(True)
o32 = O.render(c32, sw, fw, ro, rd, ts, te, bg, cd, c2w, **rck)
for k, _ in KEYS:
    gh = torch.autograd.grad((out[k] * proj[k].to(dev)).sum(), c, retain_graph=True)[0].cpu()
    go = torch.autograd.grad((o32[k] * proj[k]).sum(), c32, retain_graph=True)[0]
    fo = (out[k].detach().cpu() - o32[k].detach()).abs().max().item()
    print(f"  {k:20s} d/d planes vs fp32 oracle {rel(gh, go):.2e}   forward max abs diff {fo:.2e}")
w_h, w_o = out["weights"].detach().cpu().reshape(-1), o32["weights"].detach().reshape(-1)
idx = torch.argsort((w_h - w_o).abs(), descending=True)[:5]
print("  largest weight differences:", [(int(i), float(w_h[i]), float(w_o[i])) for i in idx])


def total(outd, pr, terms):
    loss = 0.0
    if "keys" in terms:
        for k, p_ in pr.items():
            loss = loss + (outd[k] * p_).sum()
    if "sparsity" in terms:
        loss = loss + (outd["opacity"] ** 2 + 0.01).sqrt().mean()
    if "eikonal" in terms:
        loss = loss + ((torch.linalg.norm(outd["sdf_grad"], ord=2, dim=-1) - 1.0) ** 2).mean()
    return loss


pd = {k: v.to(dev) for k, v in proj.items()}
for terms in (("keys",), ("sparsity",), ("eikonal",), ("keys", "sparsity"), ("keys", "eikonal"), ("keys", "sparsity", "eikonal")):
    gh = torch.autograd.grad(total(out, pd, terms), c, retain_graph=True)[0].cpu()
    go = torch.autograd.grad(total(o32, proj, terms), c32, retain_graph=True)[0]
    print(f"  loss terms {terms}: d/d planes vs fp32 oracle {rel(gh, go):.2e}   |g| {float(go.norm()):.3e}")
