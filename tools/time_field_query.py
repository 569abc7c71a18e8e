"""Dev tool: latency of the implicit-field query (tt_query_field: sdf + deformation head) on a 160^3 grid and of the
per-point query with normals + features on 300 k points (BASELINE configs[4], the mesh export path)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import triplaneturbo_amd as tt

from triplaneturbo_amd import _lib
if os.environ.get("TT_LIB_VARIANT"):  # dev A/B of an experiment build (tools/build_variants.py)
    _lib.use_variant(os.environ["TT_LIB_VARIANT"])
dev = torch.device("cuda", 0)
torch.manual_seed(0)
g = tt.find("few-step-triplane-dual-stable-diffusion")({"isosurface_deformable_grid": True}).to(dev)
if os.environ.get("TT_PRECISION"):
    g.precision = os.environ["TT_PRECISION"]
cache = (torch.randn(1, 6, 32, 256, 256) * 0.5).to(dev)
lin = torch.linspace(-1.0, 1.0, 160, device=dev)
grid = torch.stack(torch.meshgrid(lin, lin, lin, indexing="ij"), -1).reshape(1, -1, 3)
pts = torch.rand(1, 300_000, 3, device=dev) * 2 - 1


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


with torch.no_grad():
    print("forward_field 160^3 (4.1 M points): %.3f ms" % timed(lambda: g.forward_field(grid, cache)))
    print("geometry.forward 300 k points, normals + features: %.3f ms" % timed(lambda: g(pts, cache, output_normal=True)))
