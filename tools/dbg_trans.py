import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from triplaneturbo_amd import ops
inp = bench.make_inputs(0, torch.device("cuda", 0))
rc = ops.RenderConfig()
ro, rd = inp["ro"].reshape(-1, 3), inp["rd"].reshape(-1, 3)
packed = ops.planes_pack(inp["cache"].detach())
sw, fw = [w.detach() for w in inp["sw"]], [w.detach() for w in inp["fw"]]
r = ops.render_forward_raw(packed, sw, fw, ro, rd, inp["ts"], inp["te"], 65536, rc, image_w=256)
tr = r["trans"].view(-1, 128); w = r["weights"].view(-1, 128)
print("T0==1:", bool((tr[:, 0] == 1).all()), "nonfinite", (~torch.isfinite(tr)).sum().item())
bad = (tr[:, 1:] > tr[:, :-1] * (1 + 4e-7))
print("violations", bad.sum().item())
idx = bad.nonzero()[:5]
for ray, s in idx.tolist():
    print(ray, s, tr[ray, max(0,s-2):s+3].tolist(), w[ray, max(0,s-2):s+3].tolist())
