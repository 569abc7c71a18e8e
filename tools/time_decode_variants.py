"""Dev tool: time of the four decode-kernel variants (sdf / + normal chain / + texture half / all) at the headline size."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
from triplaneturbo_amd import ops
dev = torch.device("cuda", 0)
inp = bench.make_inputs(0, 1, dev, 1)
rc = ops.RenderConfig()
packed = ops.planes_pack(inp["cache"].detach())
ro, rd = inp["ro"].reshape(-1, 3).contiguous(), inp["rd"].reshape(-1, 3).contiguous()
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
sw = [w.detach() for w in inp["sw"]]; fw = [w.detach() for w in inp["fw"]]
for n, t in ((False, False), (True, False), (False, True), (True, True)):
    ms = timed(lambda: ops.decode_rays(packed, sw, fw, ro, rd, inp["ts"], inp["te"], 65536, rc, need_normal=n, need_features=t, image_w=256))
    print("decode_rays normal=%s tex=%s: %.3f ms" % (n, t, ms))
