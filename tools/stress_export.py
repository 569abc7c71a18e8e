"""Dev tool: repeat the per-point decode on identical inputs (300 k incoherent points) and report run-to-run differences
(must be none), then show per-lane errors against the oracle for tiles that differ.  Found the stale-lane-mask corruption
described in csrc/tt_device.h (corners_setup) and tests/test_gpu_determinism.py."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cpu_ref as O
from triplaneturbo_amd import ops

dev = torch.device("cuda", 0)
gen = torch.Generator().manual_seed(8)
cache = (torch.randn(1, 6, 32, 256, 256, generator=gen) * 0.5).to(dev)
sw = [w.to(dev) for w in O.init_mlp_weights([32, 64, 64, 1], gen)]
fw = [w.to(dev) for w in O.init_mlp_weights([96, 64, 64, 3], gen)]
n_v = 300_000
v = torch.nn.functional.normalize(torch.randn(n_v, 3, generator=gen), dim=-1) * (0.5 + 0.05 * torch.randn(n_v, 1, generator=gen))
v = v.to(dev)[None]
packed = ops.planes_pack(cache)
for need_n in (False, True):
    ref = None
    for it in range(30):
        sdf, grad, feat = ops.query_points(packed, sw, fw, v, need_normal=need_n, need_features=True)
        torch.cuda.synchronize()
        cur = (sdf.clone(), feat.clone())
        if ref is None:
            ref = cur
            continue
        for name, a, b in (("sdf", ref[0], cur[0]), ("feat", ref[1], cur[1])):
            bad = (a != b).any(dim=-1).nonzero().flatten()
            if bad.numel():
                print(f"need_n={need_n} it={it} {name}: {bad.numel()} points differ; first {bad[:16].tolist()} "
                      f"tiles {sorted(set((bad // 32).tolist()))[:8]} lanes {sorted(set((bad % 32).tolist()))}")
print("done")

# detail: which run is wrong, and how
need_n = False
outs = []
for it in range(6):
    sdf, grad, feat = ops.query_points(packed, sw, fw, v, need_normal=need_n, need_features=True)
    torch.cuda.synchronize()
    outs.append(feat.clone())
st = torch.stack(outs)  # (6, N, 3)
med = st.median(dim=0).values
for it in range(6):
    bad = (st[it] != med).any(dim=-1).nonzero().flatten()
    tiles = sorted(set((bad // 32).tolist()))
    print("run", it, "bad tiles", tiles)
    for t in tiles[:2]:
        sl = slice(t * 32, t * 32 + 32)
        want = O.geometry_forward(v[:, sl].cpu(), cache.cpu(), [w.cpu() for w in sw], [w.cpu() for w in fw], output_normal=False)["features"]
        print(" tile", t, "err_run", (st[it][sl].cpu() - want).abs().max(dim=-1).values.tolist())
        print(" tile", t, "err_med", (med[sl].cpu() - want).abs().max(dim=-1).values.max().item())
