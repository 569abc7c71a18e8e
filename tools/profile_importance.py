"""Dev tool: the bench's `importance_193` secondary workload alone (configs[1] with the reference's 128 + 64 sampler) for a
kernel trace: bash tools/kstats.sh gpurun_out/imp python tools/profile_importance.py [steps] [tile_sb]; with the work
accounting of the three decode kernels printed at the end."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import triplaneturbo_amd as tt  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda", 0)
inp = bench.make_inputs(0, 1, dev, 1)
torch.manual_seed(0)
geo = tt.find("few-step-triplane-dual-stable-diffusion")({}).to(dev)
with torch.no_grad():
    for dst, src in zip(list(geo.sdf_network.weights()) + list(geo.feature_network.weights()), inp["sw"] + inp["fw"]):
        dst.copy_(src)
base = dict(estimator="importance", trainable_variance=False, learned_variance_init=0.4605, num_samples_per_ray=64,
            num_samples_per_ray_importance=128, near_plane=0.1, far_plane=4.0, randomized=True)
r1 = tt.find("generative-space-sdf-volume-renderer")(base, geometry=geo, material=tt.find("no-material")({}),
                                                     background=tt.find("solid-color-background")({})).to(dev)
r1.train()
if len(sys.argv) > 2:
    r1.tile_sb_importance = int(sys.argv[2])
cache = inp["cache"][:1].detach().clone().requires_grad_(True)
kw = dict(space_cache=cache, text_embed=torch.zeros(1, 77, 1024), camera_distances=inp["cd"][:1], c2w=inp["c2w"][:1])
stats = torch.zeros((3, 4), dtype=torch.int64, device=dev)
if hasattr(r1, "stats"):
    r1.stats = stats


def step():
    out = r1(inp["ro"][:1], inp["rd"][:1], None, inp["bg"], **kw)
    loss = bench.loss_fn(out, {k: v[:1] for k, v in inp["proj"].items()}, fused_eikonal=True)
    for p_ in [cache] + list(geo.parameters()):
        p_.grad = None
    loss.backward()


for _ in range(2):
    step()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
ev[0].record()
for k in range(steps):
    step()
    ev[k + 1].record()
torch.cuda.synchronize()
ts = sorted(ev[k].elapsed_time(ev[k + 1]) for k in range(steps))
print("importance_193 ms/step median %.3f min %.3f" % (ts[len(ts) // 2], ts[0]))
from triplaneturbo_amd import ops  # noqa: E402
t = ops.KernelTimer()
ops.set_kernel_timer(t)
for _ in range(steps):
    step()
torch.cuda.synchronize()
ops.set_kernel_timer(None)
print({k: round(v[0] * v[1] / steps, 3) for k, v in t.summary().items()})
