"""Dev tool: per-kernel timings of one C2 step, optionally with profiling-only ablation flags.
usage: python tools/profile_kernels.py [steps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from triplaneturbo_amd import _lib  # noqa: E402

_lib.use_tuning_build()  # the -DTT_TUNING variant: honours TT_DEBUG_FLAGS & co (the product library does not)
from triplaneturbo_amd import functional, ops  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda", 0)
inp = bench.make_inputs(0, 1, dev, 1)
rc = ops.RenderConfig()
params = [inp["cache"]] + inp["sw"] + inp["fw"]


def step():
    for t in params:
        t.grad = None
    out = functional.volume_render(inp["cache"], inp["sw"], inp["fw"], inp["ro"], inp["rd"], inp["ts"], inp["te"],
                                   inp["bg"], inp["cd"], inp["c2w"], rc, training=True)
    bench.loss_fn(out, inp["proj"]).backward()


variants = (("full", "0"), ("no_scatter", "0x100"), ("no_wgrad", "0x200"), ("no_scatter_no_wgrad", "0x300"))
if len(sys.argv) > 2:  # explicit list of flag values, e.g. 0 0x2000 0x4000
    variants = tuple((f, f) for f in sys.argv[2:])
for name, flags in variants:
    os.environ["TT_DEBUG_FLAGS"] = flags
    step()
    t = ops.KernelTimer()
    ops.set_kernel_timer(t)
    torch.cuda.synchronize()
    for _ in range(steps):
        step()
    ops.set_kernel_timer(None)
    print(name, {k: round(v[0], 3) for k, v in t.summary().items()}, flush=True)
