"""Dev tool: what the opt-in backward skip (RenderConfig.skip_eps_tex / renderer.grad_skip_eps_tex) buys and costs.
Bench scene (configs[1], G6 loss) and the reference training shapes (PatchRenderer, importance sampling): ms per step
and the induced relative error of the texture-plane / feature-net gradients for thresholds relative to max |cbar|.
usage: python tools/time_skip.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import triplaneturbo_amd as tt  # noqa: E402
from triplaneturbo_amd import functional, ops, synthetic  # noqa: E402

dev = torch.device("cuda", 0)


def timeit(step, n=10):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm()).item()


# ---- bench scene ----
inp = bench.make_inputs(0, 1, dev, 1)
params = [inp["cache"]] + inp["sw"] + inp["fw"]


def bench_step(rc):
    for t in params:
        t.grad = None
    out = functional.volume_render(inp["cache"], inp["sw"], inp["fw"], inp["ro"], inp["rd"], inp["ts"], inp["te"],
                                   inp["bg"], inp["cd"], inp["c2w"], rc, training=True)
    bench.loss_fn(out, inp["proj"], fused_eikonal=True).backward()
    return out


out = bench_step(ops.RenderConfig())
s = torch.sigmoid(out["features"].detach())
grgb = inp["proj"]["comp_rgb"].reshape(-1, 3).repeat_interleave(128, dim=0)
cb = (out["weights"].detach() * grgb * 1.002 * s * (1 - s)).abs().sum(-1)
cmax = cb.max().item()
ref = [t.grad.clone() for t in params]
print(f"bench scene: max |cbar|_1 = {cmax:.3e}")
for frac in (0.0, 1e-6, 1e-5, 1e-4, 1e-3):
    rc = ops.RenderConfig(skip_eps_tex=frac * cmax)
    ms = timeit(lambda: bench_step(rc))
    bench_step(rc)
    g = [t.grad for t in params]
    print(f"  skip_eps_tex = {frac:g} x max: {ms:.3f} ms/step, samples below eps {float((cb <= frac * cmax).double().mean()):.3f}, "
          f"rel err planes(tex) {rel(g[0][:, 3:], ref[0][:, 3:]):.2e} v1 {rel(g[4], ref[4]):.2e} v2 {rel(g[5], ref[5]):.2e} "
          f"v3 {rel(g[6], ref[6]):.2e}")

# ---- reference training shapes ----
torch.manual_seed(0)
geo = tt.find("few-step-triplane-dual-stable-diffusion")({}).to(dev)
base = dict(estimator="importance", trainable_variance=False, learned_variance_init=0.4605, num_samples_per_ray=64,
            num_samples_per_ray_importance=128, near_plane=0.1, far_plane=4.0, randomized=False)
r = tt.find("patch-renderer")({"patch_size": 40, "global_downsample": 3,
                               "base_renderer_type": "generative-space-sdf-volume-renderer", "base_renderer": base},
                              geometry=geo, material=tt.find("no-material")({}),
                              background=tt.find("solid-color-background")({})).to(dev)
r.train()
gen = torch.Generator().manual_seed(1)
cache = (torch.randn(2, 6, 32, 256, 256, generator=gen) * 0.5).to(dev).requires_grad_(True)
ro, rd, c2w, cd = synthetic.make_cameras(8, 128, 128)
kw = dict(space_cache=cache, text_embed=torch.zeros(2, 77, 1024), camera_distances=cd.to(dev), c2w=c2w.to(dev))
ro, rd, bg = ro.to(dev), rd.to(dev), torch.ones(3, device=dev)
gp = [cache] + list(geo.parameters())


def train_step():
    torch.manual_seed(3)  # same random patch every step
    o = r(ro, rd, None, bg, **kw)
    loss = o["comp_rgb"].mean() + (o["opacity"] ** 2 + 0.01).sqrt().mean() + ops.eikonal_loss(o["sdf_grad"])
    for p_ in gp:
        p_.grad = None
    loss.backward()


train_step()
ref = [p_.grad.clone() for p_ in gp]
scale = 1.0 / (8 * 128 * 128 * 3) * 0.2505  # d mean / d comp_rgb x max of 1.002 s (1 - s), weights <= 1
print(f"training shapes (2 prompts x 4 views, 42^2 + 40^2 rays, 193 samples): |cbar|_1 <= {3 * scale:.3e}")
for frac in (0.0, 1e-5, 1e-4, 1e-3, 1e-2):
    r.base_renderer.grad_skip_eps_tex = frac * 3 * scale
    ms = timeit(train_step)
    train_step()
    g = [p_.grad for p_ in gp]
    names = [n for n, _ in geo.named_parameters()]
    errs = {n: rel(a, b) for n, a, b in zip(["planes_tex"] + names, [g[0][:, 3:]] + g[1:], [ref[0][:, 3:]] + ref[1:])
            if "feature" in n or n == "planes_tex"}
    print(f"  grad_skip_eps_tex = {frac:g} x bound: {ms:.3f} ms/step, rel err " +
          ", ".join(f"{k} {v:.2e}" for k, v in errs.items()))
