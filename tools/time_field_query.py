"""Dev tool: time the isosurface grid query of the inference path (mesh_exporter.py:78-105: forward_field on a 160^3
grid, then vertex colouring = geometry.export on ~300k points).   usage: python tools/time_field_query.py [res]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import triplaneturbo_amd as tt  # noqa: E402

res = int(sys.argv[1]) if len(sys.argv) > 1 else 160
dev = torch.device("cuda", 0)
torch.manual_seed(0)
g = tt.find("few-step-triplane-dual-stable-diffusion")({"isosurface_deformable_grid": True}).to(dev)
g.eval()
cache = (torch.randn(1, 6, 32, 256, 256) * 0.5).to(dev)
lin = torch.linspace(-1.0, 1.0, res, device=dev)
pts = torch.stack(torch.meshgrid(lin, lin, lin, indexing="ij"), -1).reshape(1, -1, 3).contiguous()
verts = (torch.rand(1, 300000, 3, device=dev) * 1.6 - 0.8)


def timeit(fn, n=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    t_field = timeit(lambda: g.forward_field(pts, cache))
    t_full = timeit(lambda: g(pts[:, :2000000], cache, output_normal=True))
    t_export = timeit(lambda: g.export(verts, cache))
print(f"forward_field {res}^3 = {pts.shape[1] / 1e6:.2f} M points: {t_field:.2f} ms "
      f"({pts.shape[1] / t_field / 1e3:.0f} M points/s); forward(+normal+features) 2 M points: {t_full:.2f} ms; "
      f"export 300k vertices: {t_export:.3f} ms", flush=True)
