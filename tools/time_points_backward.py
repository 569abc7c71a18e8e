"""Dev tool: the raster path's per-pixel geometry decode with d/d points: geometry.forward on N points with normals
and features, backward of a scalar loss to the planes, the weights AND the query points (tt_points_bwd_geo / _tex / _x).
usage: python tools/time_points_backward.py [n_points]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import triplaneturbo_amd as tt
from triplaneturbo_amd import _lib, ops
if os.environ.get("TT_LIB_VARIANT"):  # dev A/B of an experiment build (tools/build_variants.py)
    _lib.use_variant(os.environ["TT_LIB_VARIANT"])
dev = torch.device("cuda", 0)
torch.manual_seed(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
g = tt.find("few-step-triplane-dual-stable-diffusion")({}).to(dev)
if os.environ.get("TT_PRECISION"):
    g.precision = os.environ["TT_PRECISION"]
cache = (torch.randn(1, 6, 32, 256, 256) * 0.5).to(dev).requires_grad_(True)
pts = (torch.rand(1, n, 3, device=dev) * 2 - 1).requires_grad_(True)
up = {k: torch.randn(1, n, c, device=dev) for k, c in (("sdf", 1), ("normal", 3), ("features", 3))}


def step():
    for t in [cache, pts] + list(g.parameters()):
        t.grad = None
    out = g(pts, cache, output_normal=True)
    sum((out[k] * up[k]).sum() for k in up).backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    step()
e1.record()
torch.cuda.synchronize()
t = ops.KernelTimer()
ops.set_kernel_timer(t)
for _ in range(20):
    step()
torch.cuda.synchronize()
ops.set_kernel_timer(None)
per_step = {k: round(v[0] * v[1] / 20, 3) for k, v in t.summary().items()}
print("geometry.forward + backward (planes, weights, points), %d points: %.3f ms per step; entry points %s"
      % (n, e0.elapsed_time(e1) / 20, per_step))
