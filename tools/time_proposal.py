"""Dev tool: the sampler's proposal pass alone -- tt_decode_rays, sdf head only -- at 256 x 256 rays x 128 samples (and at the
training shape), per precision mode: ms per launch and a checksum of the output (A/B of library variants: the checksums must
agree bit for bit).  usage: python tools/time_proposal.py [--variant NAME]"""
import hashlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from triplaneturbo_amd import _lib  # noqa: E402
args = sys.argv[1:]
if args and args[0] == "--variant":
    _lib.use_variant(args[1])
from triplaneturbo_amd import ops, synthetic  # noqa: E402

dev = "cuda"
g = torch.Generator().manual_seed(0)
cache = (torch.randn(2, 6, 32, 256, 256, generator=g) * 0.5).to(dev)
sw = [w.to(dev) for w in synthetic.init_mlp_weights([32, 64, 64, 1], g)]
packed = ops.planes_pack(cache)
for name, P, NV, Hh, Ww, S, sb in (("configs[1] 256x256 rays x 128", 1, 1, 256, 256, 128, 0), ("training 2x4 views 42x42 x 128", 2, 4, 42, 42, 128, 8)):
    ro, rd, _, _ = synthetic.make_cameras(P * NV, Hh, Ww)
    ts, te = synthetic.uniform_intervals(P * NV * Hh * Ww, S, 0.1, 4.0)
    ro, rd, ts, te = ro.reshape(-1, 3).to(dev), rd.reshape(-1, 3).to(dev), ts.to(dev), te.to(dev)
    chunks = [int(c) for c in os.environ.get("TT_CHUNKS", "0").split(",")]  # samples per work item (0 = automatic)
    for prec, chunk in [(pr, c) for pr in os.environ.get("TT_PRECISIONS", "split3,f32,split2").split(",") for c in chunks]:
        rc = ops.RenderConfig(precision=prec, tile_sb=sb, tile_chunk=chunk)
        run = lambda: ops.decode_rays(packed[:P], sw, None, ro, rd, ts, te, Hh * Ww, rc, need_normal=False, need_features=False, image_w=Ww)  # noqa: E731
        sdf = run()[0]
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50):
            run()
        b.record()
        torch.cuda.synchronize()
        print(f"{name:34s} {prec:7s} chunk {chunk:3d}  {a.elapsed_time(b) / 50:.4f} ms   sha {hashlib.sha256(sdf.cpu().numpy().tobytes()).hexdigest()[:16]}")
