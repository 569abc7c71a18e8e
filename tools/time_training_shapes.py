"""Dev tool: time one PatchRenderer forward+backward at the reference TRAINING shapes (2 prompts x 4 views, 42^2
global + 40^2 patch rays, importance sampling 128 + 64 -> 193 samples, planes 256^2) with the per-entry-point
breakdown (HIP events) next to the wall clock.     usage: python tools/time_training_shapes.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import triplaneturbo_amd as tt  # noqa: E402
from triplaneturbo_amd import _lib, ops, synthetic  # noqa: E402

if os.environ.get("TT_TUNING_BUILD"):  # the -DTT_TUNING library: honours TT_DEBUG_FLAGS (ablations)
    _lib.use_tuning_build()
if os.environ.get("TT_LIB_VARIANT"):  # dev A/B of an experiment build (tools/build_variants.py)
    _lib.use_variant(os.environ["TT_LIB_VARIANT"])

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
P = int(sys.argv[2]) if len(sys.argv) > 2 else 2
NV = int(sys.argv[3]) if len(sys.argv) > 3 else 4
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
if os.environ.get("TT_SB_IMPORTANCE"):  # sweep of the tile shape used under importance sampling
    for m in r.modules():
        if hasattr(m, "tile_sb_importance"):
            m.tile_sb_importance = int(os.environ["TT_SB_IMPORTANCE"])
if os.environ.get("TT_SB_GLOBAL"):
    r.tile_sb_global = int(os.environ["TT_SB_GLOBAL"])
if os.environ.get("TT_SB_PATCH"):
    r.tile_sb_patch = int(os.environ["TT_SB_PATCH"])
if os.environ.get("TT_GRAD_COPIES") or os.environ.get("TT_TILE_CHUNK"):
    # privatised plane-gradient copies (copy = workgroup id % copies: 8 = one per XCD) / samples of a ray block per work item
    for m in r.modules():
        if hasattr(m, "_render_config"):
            def _rc(orig=m._render_config):
                rc = orig()
                rc.grad_copies = int(os.environ.get("TT_GRAD_COPIES", rc.grad_copies))
                rc.tile_chunk = int(os.environ.get("TT_TILE_CHUNK", rc.tile_chunk))
                return rc
            m._render_config = _rc
gen = torch.Generator().manual_seed(1)
cache = (torch.randn(P, 6, 32, 256, 256, generator=gen) * 0.5).to(dev).requires_grad_(True)
ro, rd, c2w, cd = synthetic.make_cameras(P * NV, 128, 128)
kw = dict(space_cache=cache, text_embed=torch.zeros(P, 77, 1024), camera_distances=cd.to(dev), c2w=c2w.to(dev))
ro, rd = ro.to(dev), rd.to(dev)
bg = torch.ones(3, device=dev)


# TT_LOSS_KEYS="comp_rgb,opacity,depth,comp_normal_cam_vis": seeded projections of those outputs (bench.py --config 2's loss)
# instead of comp_rgb.mean(); the sparsity and eikonal terms stay
_keys = [k for k in os.environ.get("TT_LOSS_KEYS", "").split(",") if k]
_gen = torch.Generator().manual_seed(5)
_proj = {k: torch.randn(P * NV, 128, 128, {"comp_rgb": 3, "opacity": 1, "depth": 1, "comp_normal_cam_vis": 3}[k],
                        generator=_gen).to(dev) * float(os.environ.get("TT_LOSS_SCALE", "1")) for k in _keys}


def step():
    out = r(ro, rd, None, bg, **kw)
    loss = (sum((out[k] * v).sum() for k, v in _proj.items()) if _keys else out["comp_rgb"].mean()) + \
        (out["opacity"] ** 2 + 0.01).sqrt().mean() + ((out["sdf_grad"].norm(dim=-1) - 1) ** 2).mean()
    for p_ in [cache] + list(g.parameters()):
        p_.grad = None
    loss.backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / steps * 1e3
t = ops.KernelTimer()
ops.set_kernel_timer(t)
for _ in range(steps):
    step()
torch.cuda.synchronize()
ops.set_kernel_timer(None)
per_step = {k: round(v[0] * v[1] / steps, 3) for k, v in t.summary().items()}  # avg ms x launches / steps
print(f"training shapes ({P} prompts x {NV} views): wall {wall:.2f} ms/step; HIP entry points (ms/step): {per_step} sum "
      f"{sum(per_step.values()):.2f}", flush=True)
