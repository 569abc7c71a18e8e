"""Dev tool: where does the default (split-fp16) path of one fuzz seed leave the exact_f32 path?  Per-sample forward
outputs and per-output-key gradients.  usage: python tools/debug_seed.py SEED"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import cpu_ref as O  # noqa: E402
import test_gpu_fuzz as F  # noqa: E402
from test_gpu_backward import KEYS  # noqa: E402
from parity import rel  # noqa: E402
from triplaneturbo_amd import _lib  # noqa: E402

if os.environ.get("TT_LIB_VARIANT"):
    _lib.use_variant(os.environ["TT_LIB_VARIANT"])
from triplaneturbo_amd import functional, ops  # noqa: E402

seed = int(sys.argv[1])
P, n_view, R, Hh, Ww, S, rck, knobs, near, far, jittered = F._case(seed)
g = torch.Generator().manual_seed(seed)
cache = torch.randn(P, 6, 32, R, R, generator=g) * 0.5
sw = O.init_mlp_weights([32, 64, 64, 1], g)
fw = O.init_mlp_weights([96, 64, 64, 3], g)
ro, rd, c2w, cd = O.make_cameras(P * n_view, Hh, Ww)
n_rays = P * n_view * Hh * Ww
ts, te = O.uniform_intervals(n_rays, S, near, far)
if jittered:
    edges = torch.cat([ts[:, :1], te], dim=1)
    w = (far - near) / S
    edges[:, 1:-1] += (torch.rand(n_rays, S - 1, generator=g) - 0.5) * 0.9 * w
    ts, te = edges[:, :-1].contiguous(), edges[:, 1:].contiguous()
bg = torch.rand(3, generator=g)
proj = {n: torch.randn(P * n_view, Hh, Ww, c, generator=g) for n, c in KEYS}
if not os.environ.get("TT_FUZZ_NO_KINK_MASK"):  # as the fuzz test does
    from parity import kink_free_rays  # noqa: E402
    keep = kink_free_rays(cache, sw, fw, ro, rd, ts, te, n_view)
    print("kink-free rays", int(keep.sum()), "of", keep.numel())
    proj = {n: v * keep.view(P * n_view, Hh, Ww, 1).to(v.dtype) for n, v in proj.items()}
dev = "cuda"


def hip(precision, keys=None):
    c = cache.to(dev).requires_grad_(True)
    sws = [w.to(dev).requires_grad_(True) for w in sw]
    fws = [w.to(dev).requires_grad_(True) for w in fw]
    rc = ops.RenderConfig(**dict(rck, precision=precision, **{k: v for k, v in knobs.items() if k != "precision"}))
    out = functional.volume_render(c, sws, fws, ro.to(dev), rd.to(dev), ts.to(dev), te.to(dev), bg.to(dev), cd.to(dev),
                                   c2w.to(dev), rc, training=True)
    res = {}
    for k in ("sdf", "sdf_grad", "features", "weights", "comp_rgb", "opacity", "depth", "comp_normal"):
        res[k] = out[k].detach().double().cpu()
    grads = {}
    for k, _ in (KEYS if keys is None else keys):
        gs = torch.autograd.grad((out[k] * proj[k].to(dev)).sum(), [c] + sws + fws, retain_graph=True, allow_unused=True)
        grads[k] = [None if t is None else t.detach().double().cpu() for t in gs]
    # per-sample upstream probes: gradient of a loss on sdf / sdf_grad alone
    for name, t in (("sum(sdf)", out["sdf"].sum()), ("sum(sdf_grad^2)", (out["sdf_grad"] ** 2).sum())):
        gs = torch.autograd.grad(t, [c] + sws + fws, retain_graph=True, allow_unused=True)
        grads[name] = [None if x is None else x.detach().double().cpu() for x in gs]
    return res, grads


def oracle(dt):
    c = cache.to(dt).requires_grad_(True)
    sws = [w.to(dt).requires_grad_(True) for w in sw]
    fws = [w.to(dt).requires_grad_(True) for w in fw]
    out = O.render(c, sws, fws, ro.to(dt), rd.to(dt), ts.to(dt), te.to(dt), bg.to(dt), cd.to(dt), c2w.to(dt), **rck)
    res = {k: out[k].detach().double() for k in ("sdf", "sdf_grad", "features", "weights", "comp_rgb", "opacity", "depth", "comp_normal") if k in out}
    grads = {}
    for k, _ in KEYS:
        gs = torch.autograd.grad((out[k] * proj[k].to(dt)).sum(), [c] + sws + fws, retain_graph=True, allow_unused=True)
        grads[k] = [None if t is None else t.detach().double() for t in gs]
    for name, t in (("sum(sdf)", out["sdf"].sum()), ("sum(sdf_grad^2)", (out["sdf_grad"] ** 2).sum())):
        gs = torch.autograd.grad(t, [c] + sws + fws, retain_graph=True, allow_unused=True)
        grads[name] = [None if x is None else x.detach().double() for x in gs]
    return res, grads


rd_, gd = hip(os.environ.get("TT_DEBUG_MODE", "split3"))  # "default" columns below = this mode
rx, gx = hip("f32")
r64, g64 = oracle(torch.float64)
r32, g32 = oracle(torch.float32)
print("seed", seed, rck, "P", P, "views", n_view, "R", R, Hh, Ww, "S", S)
print("forward outputs: relative error against fp64 (norm-wise) and worst element / full scale")
for k in rd_:
    if k not in r64:
        continue
    a, b, c_, d = rd_[k].reshape(-1), rx[k].reshape(-1), r32[k].reshape(-1), r64[k].reshape(-1)
    fs = d.abs().max().clamp_min(1e-30)
    print(f"  {k:12s} default {rel(a, d):.2e} (max {((a - d).abs().max() / fs):.2e})  exact {rel(b, d):.2e} (max {((b - d).abs().max() / fs):.2e})"
          f"  fp32 oracle {rel(c_, d):.2e} (max {((c_ - d).abs().max() / fs):.2e})   |x|max {fs:.3e}")
names = ["space_cache", "sdf.w1", "sdf.w2", "sdf.w3", "feat.v1", "feat.v2", "feat.v3"]
print("gradients of <out[k], proj[k]> per key, error against fp64: default / exact / fp32 oracle")
for k in gd:
    row = []
    for i, n in enumerate(names):
        if g64[k][i] is None or float(g64[k][i].abs().max()) == 0 or gd[k][i] is None:
            continue
        row.append(f"{n} {rel(gd[k][i], g64[k][i]):.1e}/{rel(gx[k][i], g64[k][i]):.1e}/{rel(g32[k][i], g64[k][i]):.1e}")
    print(f"  {k:20s} " + "  ".join(row))

# ---- ReLU kinks: where the default path's normal is off, is a hidden pre-activation within rounding distance of zero? ----
with torch.no_grad():
    tm = ((ts + te) * 0.5).double()
    pts = ro.reshape(-1, 1, 3).double() + rd.reshape(-1, 1, 3).double() * tm[..., None]          # (n_rays, S, 3)
    B = P * n_view
    pts = pts.reshape(B, Hh * Ww * S, 3)
og = O.geometry_forward(pts, cache.double().repeat_interleave(n_view, 0), [w.double() for w in sw], [w.double() for w in fw])
f = og["enc_geo"].detach()
W1, W2 = sw[0].double(), sw[1].double()
h1p = f @ W1.T
h1 = torch.relu(h1p)
h2p = h1 @ W2.T
s1 = f.abs() @ W1.abs().T
s2 = h1.abs() @ W2.abs().T
m1 = (h1p.abs() / s1.clamp_min(1e-300))
m2 = (h2p.abs() / s2.clamp_min(1e-300))
m1[s1 == 0] = 1.0
m2[s2 == 0] = 1.0
near1, near2 = m1.min(dim=1).values, m2.min(dim=1).values
err_d = (rd_["sdf_grad"].reshape(-1, 3) - r64["sdf_grad"].reshape(-1, 3)).norm(dim=1)
err_x = (rx["sdf_grad"].reshape(-1, 3) - r64["sdf_grad"].reshape(-1, 3)).norm(dim=1)
print("samples with the largest |sdf_grad(default) - sdf_grad(fp64)|: error default / exact; smallest |pre-activation| / sum|w||x| of layer 1 / 2 (fp64)")
for idx in torch.argsort(err_d, descending=True)[:6].tolist():
    print(f"  sample {idx}: {err_d[idx]:.2e} / {err_x[idx]:.2e}   layer1 {near1[idx]:.2e}  layer2 {near2[idx]:.2e}")
print("all samples: fraction with a pre-activation closer than 2^-20 / 2^-22 / 2^-24 of its sum|w||x|:",
      [float(((near1 < t) | (near2 < t)).double().mean()) for t in (2.0 ** -20, 2.0 ** -22, 2.0 ** -24)], "of", err_d.numel())

# ---- discrete events of the march: alpha at a clip boundary, cos(theta) at the relu kink ----
with torch.no_grad():
    w_d, w_x, w_32, w_64 = (r["weights"].reshape(-1) for r in (rd_, rx, r32, r64))
    dw = (w_d - w_32).abs()
    print("samples with the largest |weights(mode) - weights(fp32 oracle)|: mode / f32-mode / fp32 oracle / fp64; sdf of each")
    for idx in torch.argsort(dw, descending=True)[:8].tolist():
        print(f"  sample {idx} (ray {idx // S}, s {idx % S}): w {w_d[idx]:.6e} / {w_x[idx]:.6e} / {w_32[idx]:.6e} / {w_64[idx]:.6e}   "
              f"sdf {rd_['sdf'].reshape(-1)[idx]:.8e} / {rx['sdf'].reshape(-1)[idx]:.8e} / {r32['sdf'].reshape(-1)[idx]:.8e} / {r64['sdf'].reshape(-1)[idx]:.8e}")
    # fp64 alpha terms per sample
    n64 = torch.nn.functional.normalize(r64["sdf_grad"].reshape(-1, 3), dim=-1)
    t_dirs = rd.reshape(-1, 3).double().repeat_interleave(S, 0)
    cosv = (t_dirs * n64).sum(-1)
    dists = (te - ts).reshape(-1).double()
    k = rck["inv_std"]
    car = rck["cos_anneal_ratio"]
    ic = -(torch.relu(-cosv * 0.5 + 0.5) * (1 - car) + torch.relu(-cosv) * car)
    sd = r64["sdf"].reshape(-1)
    prev, nxt = torch.sigmoid((sd - ic * dists * 0.5) * k), torch.sigmoid((sd + ic * dists * 0.5) * k)
    ratio = (prev - nxt + 1e-5) / (prev + 1e-5)
    print("fp64: samples whose alpha ratio is within 1e-5 of a clip bound (0 / 1):", int(((ratio.abs() < 1e-5) | ((ratio - 1).abs() < 1e-5)).sum()),
          "  |cos| < 1e-5:", int((cosv.abs() < 1e-5).sum()), " of", ratio.numel())
    for idx in torch.argsort(dw, descending=True)[:4].tolist():
        print(f"  sample {idx}: cos {cosv[idx]:.3e} ratio {ratio[idx]:.6e} prev_cdf {prev[idx]:.3e} next_cdf {nxt[idx]:.3e}")
