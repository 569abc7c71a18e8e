"""Dev tool: wave-pair backward kernels against the one-wave-per-tile ones (TT_R_BWD_SOLO) on the same inputs."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cpu_ref as O  # noqa: E402
from triplaneturbo_amd import functional, ops  # noqa: E402

dev = "cuda"


def run(P, R, n_view, Hh, Ww, S, seed, solo, sb=0):
    g = torch.Generator().manual_seed(seed)
    cache = (torch.randn(P, 6, 32, R, R, generator=g) * 0.5).to(dev).requires_grad_(True)
    sw = [w.to(dev).requires_grad_(True) for w in O.init_mlp_weights([32, 64, 64, 1], g)]
    fw = [w.to(dev).requires_grad_(True) for w in O.init_mlp_weights([96, 64, 64, 3], g)]
    ro, rd, c2w, cd = O.make_cameras(P * n_view, Hh, Ww)
    ts, te = O.uniform_intervals(P * n_view * Hh * Ww, S, 0.3, 3.2)
    proj = {k: torch.randn(P * n_view, Hh, Ww, c, generator=g).to(dev) for k, c in (("comp_rgb", 3), ("opacity", 1), ("depth", 1))}
    rc = ops.RenderConfig(bwd_pair=not solo, tile_sb=sb)
    out = functional.volume_render(cache, sw, fw, ro.to(dev), rd.to(dev), ts.to(dev), te.to(dev), torch.ones(3, device=dev),
                                   cd.to(dev), c2w.to(dev), rc, training=True)
    loss = O.synthetic_loss(out, proj)
    return [t.detach().double().cpu() for t in torch.autograd.grad(loss, [cache] + sw + fw)]


for cfg in [(1, 32, 1, 8, 8, 32, 0), (2, 32, 2, 5, 7, 45, 2), (1, 64, 1, 24, 24, 16, 3)]:
    for sb in (0, 8):
        a = run(*cfg, solo=False, sb=sb)
        b = run(*cfg, solo=True, sb=sb)
        names = ["planes_geo", "planes_tex", "w1", "w2", "w3", "v1", "v2", "v3"]
        vals = [a[0][:, :3], a[0][:, 3:]] + a[1:]
        refs = [b[0][:, :3], b[0][:, 3:]] + b[1:]
        print(cfg, "sb", sb, {n: float((x - y).norm() / y.norm().clamp_min(1e-30)) for n, x, y in zip(names, vals, refs)})
