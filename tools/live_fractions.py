"""Dev tool: how much of the bench workload is live per SAMPLE against per 32-sample TILE (what the kernels skip today)?
(the in-bounds test here is the textbook one; the kernels' own counters are in the bench line)
geometric: >= 1 plane with an in-bounds texel; tex: additionally a non-zero rendering weight.  usage: python tools/live_fractions.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from triplaneturbo_amd import functional, ops  # noqa: E402

dev = torch.device("cuda", 0)
inp = bench.make_inputs(0, 1, dev, 1)
rc = ops.RenderConfig()
out = functional.volume_render(inp["cache"], inp["sw"], inp["fw"], inp["ro"], inp["rd"], inp["ts"], inp["te"], inp["bg"],
                               inp["cd"], inp["c2w"], rc, training=True)
ro, rd = inp["ro"].reshape(-1, 3), inp["rd"].reshape(-1, 3)
ts, te = inp["ts"], inp["te"]
n, S = ts.shape
tm = 0.5 * (ts + te)
p = ro[:, None, :] + rd[:, None, :] * tm[..., None]
R = inp["cache"].shape[-1]
inb = (p.abs() / rc.radius) < 1.0 + 1.0 / R  # per coordinate
cnt = inb.sum(-1)
geo = cnt >= 2  # a plane needs both of its coordinates
planes = (inb[..., 0] & inb[..., 1]).float() + (inb[..., 0] & inb[..., 2]).float() + (inb[..., 1] & inb[..., 2]).float()
w = out["weights"].reshape(n, S)
tex = geo & (w != 0)


def tiles(m):
    """the kernels' default tile: 4 x 4 pixels x 2 consecutive samples (tile_sb = 2)"""
    Hh = Ww = int(round(n ** 0.5))
    t = m.reshape(Hh // 4, 4, Ww // 4, 4, S // 2, 2).permute(0, 2, 4, 1, 3, 5).reshape(-1, 32)
    return t.any(-1).float().mean().item()


def run_len(m):  # live samples per ray: first / last / count
    c = m.sum(-1).float()
    return c.mean().item(), c.max().item()


print("samples", n * S, "rays", n)
print("geometric live: per sample %.4f  per tile %.4f   in-bounds plane frac %.4f" % (geo.float().mean().item(), tiles(geo), planes.mean().item() / 3))
print("weight != 0   : per sample %.4f  per tile %.4f" % ((w != 0).float().mean().item(), tiles(w != 0)))
print("weight > 1e-8 : per sample %.4f  per tile %.4f" % ((w > 1e-8).float().mean().item(), tiles(w > 1e-8)))
print("tex live      : per sample %.4f  per tile %.4f" % (tex.float().mean().item(), tiles(tex)))
print("live samples per ray (geo): mean %.1f max %.0f" % run_len(geo))
rays_live = geo.any(-1)
print("rays with any live sample %.4f" % rays_live.float().mean().item())
# compaction in ray-major order: tiles needed = ceil(live / 32) over the whole launch
for name, m in (("geo", geo), ("tex", tex)):
    print(name, "compacted tiles / visited tiles = %.4f" % ((m.sum().item() / 32) / (n * S / 32)))
