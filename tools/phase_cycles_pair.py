"""Dev tool (tuning build): cycles per phase of the wave-pair texture backward (k_decode_bwd_tex2), summed over waves."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from triplaneturbo_amd import _lib  # noqa: E402

_lib.use_tuning_build()
import bench  # noqa: E402
from triplaneturbo_amd import functional, ops  # noqa: E402

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


lib = _lib.load()
lib.tt_tuning_phase_cycles.argtypes = [ctypes.c_void_p]
buf = (ctypes.c_uint64 * 40)()
lib.tt_tuning_phase_cycles(buf)
step()
lib.tt_tuning_phase_cycles(buf)
n = 3
for _ in range(n):
    step()
lib.tt_tuning_phase_cycles(buf)
names = ["inputs + cbar + skip test", "gather half -> e fragments", "sync 1 (e fragments)", "k1 half + split + store", "sync 2 (k1)",
         "k2 half, dV3, k2bar + split + store", "sync 3 (k2bar)", "dV2 (tr operands + outer)", "k1bar half + split + store",
         "sync 4 (k1bar)", "dV1 (tr operands + outer)", "ebar of my planes", "sync 5 (before scatter)", "scatter of my planes",
         "sync 6 (after scatter), one-plane wave", "item pop + 2 syncs", "tail (flush)", "sync 6, two-plane wave"]
tot = sum(buf[:20])
waves = tot and 1
print(f"== k_decode_bwd_tex2: {tot / n / 1e6:.1f} M shader cycles summed over all waves per launch ==")
for k, nm in enumerate(names):
    print(f"{nm:40s} {100.0 * buf[k] / max(tot, 1):5.1f} %")
