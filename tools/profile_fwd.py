"""Dev tool: forward-kernel ablations (profiling-only flags)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from triplaneturbo_amd import _lib
_lib.use_tuning_build()  # -DTT_TUNING variant (honours TT_DEBUG_FLAGS)
from triplaneturbo_amd import ops
dev = torch.device("cuda", 0)
inp = bench.make_inputs(0, 1, dev, 1)
rc = ops.RenderConfig()
packed = ops.planes_pack(inp["cache"].detach())
sw = [w.detach() for w in inp["sw"]]; fw = [w.detach() for w in inp["fw"]]
ro, rd = inp["ro"].reshape(-1, 3), inp["rd"].reshape(-1, 3)
for name, flags in (("full", 0), ("no_gather", 0x400), ("no_mlp", 0x800), ("no_store", 0x1000), ("no_gather_no_mlp", 0xC00), ("none", 0x1C00)):
    os.environ["TT_DEBUG_FLAGS"] = hex(flags)
    for ps in (True, False):
        ops.render_forward_raw(packed, sw, fw, ro, rd, inp["ts"], inp["te"], 65536, rc, per_sample=ps, image_w=256)
        t = ops.KernelTimer(); ops.set_kernel_timer(t); torch.cuda.synchronize()
        for _ in range(5):
            ops.render_forward_raw(packed, sw, fw, ro, rd, inp["ts"], inp["te"], 65536, rc, per_sample=ps, image_w=256)
        ops.set_kernel_timer(None)
        print(name, "per_sample" if ps else "eval", {k: round(v[0], 3) for k, v in t.summary().items()}, flush=True)
