"""Dev tool: time the reference-training-shape step and the bench step for several grad_copies values."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import triplaneturbo_amd as tt
from triplaneturbo_amd import functional, ops, synthetic

dev = torch.device("cuda", 0)
torch.manual_seed(0)
g = tt.find("few-step-triplane-dual-stable-diffusion")({}).to(dev)
base = dict(estimator="importance", trainable_variance=False, learned_variance_init=0.4605, num_samples_per_ray=64,
            num_samples_per_ray_importance=128, near_plane=0.1, far_plane=4.0, randomized=True)
r = tt.find("patch-renderer")({"patch_size": 40, "global_downsample": 3,
                               "base_renderer_type": "generative-space-sdf-volume-renderer", "base_renderer": base},
                              geometry=g, material=tt.find("no-material")({}),
                              background=tt.find("solid-color-background")({})).to(dev)
r.train()
gen = torch.Generator().manual_seed(1)
cache = (torch.randn(2, 6, 32, 256, 256, generator=gen) * 0.5).to(dev).requires_grad_(True)
ro, rd, c2w, cd = synthetic.make_cameras(8, 128, 128)
kw = dict(space_cache=cache, text_embed=torch.zeros(2, 77, 1024), camera_distances=cd.to(dev), c2w=c2w.to(dev))
ro, rd = ro.to(dev), rd.to(dev)
inp = bench.make_inputs(0, dev)

def train_step():
    out = r(ro, rd, None, torch.ones(3, device=dev), **kw)
    loss = out["comp_rgb"].mean() + (out["opacity"] ** 2 + 0.01).sqrt().mean() + ((out["sdf_grad"].norm(dim=-1) - 1) ** 2).mean()
    for p_ in [cache] + list(g.parameters()): p_.grad = None
    loss.backward()

def bench_step(rc):
    for t in [inp["cache"]] + inp["sw"] + inp["fw"]: t.grad = None
    out = functional.volume_render(inp["cache"], inp["sw"], inp["fw"], inp["ro"], inp["rd"], inp["ts"], inp["te"],
                                   inp["bg"], inp["cd"], inp["c2w"], rc, training=True)
    bench.loss_fn(out, inp["proj"]).backward()

def timeit(fn, n=4):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

default = ops.RenderConfig.__dataclass_fields__["grad_copies"].default
for copies in [int(a) for a in sys.argv[1:]] or [1, 8]:
    ops.RenderConfig.__dataclass_fields__["grad_copies"].default = copies
    ops.RenderConfig.__init__.__defaults__ = tuple(copies if (d == default and i == 6) else d for i, d in enumerate(ops.RenderConfig.__init__.__defaults__))
    default = copies
    rc = ops.RenderConfig(grad_copies=copies)
    print(f"grad_copies={copies:3d}  training shapes {timeit(train_step):6.1f} ms   bench step {timeit(lambda: bench_step(rc)):6.1f} ms", flush=True)
