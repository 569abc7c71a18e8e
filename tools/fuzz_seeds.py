"""Dev tool: for given fuzz seeds (tests/test_gpu_fuzz.py) print every gradient's distance HIP default / HIP exact_f32 /
fp32 oracle, pairwise and from the fp64 oracle.  usage: python tools/fuzz_seeds.py 288 225 ...  | range:A:B"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import cpu_ref as O  # noqa: E402
import test_gpu_fuzz as F  # noqa: E402
from test_gpu_backward import KEYS, _hip_grads, _oracle_grads  # noqa: E402
from parity import rel  # noqa: E402
from triplaneturbo_amd import functional, ops  # noqa: E402

mods = (ops, functional)
seeds = []
for a in sys.argv[1:]:
    if a.startswith("range:"):
        _, lo, hi = a.split(":")
        seeds += list(range(int(lo), int(hi)))
    else:
        seeds.append(int(a))
names = ["space_cache", "sdf.w1", "sdf.w2", "sdf.w3", "feat.v1", "feat.v2", "feat.v3"]
worst = {}
for seed in seeds:
    P, n_view, R, Hh, Ww, S, rck, knobs, near, far, jittered = F._case(seed)
    g = torch.Generator().manual_seed(seed)
    cache = torch.randn(P, 6, 32, R, R, generator=g) * 0.5
    sw = O.init_mlp_weights([32, 64, 64, 1], g)
    fw = O.init_mlp_weights([96, 64, 64, 3], g)
    ro, rd, c2w, cd = O.make_cameras(P * n_view, Hh, Ww)
    n_rays = P * n_view * Hh * Ww
    ts, te = O.uniform_intervals(n_rays, S, near, far)
    if jittered:
        edges = torch.cat([ts[:, :1], te], dim=1)
        w = (far - near) / S
        edges[:, 1:-1] += (torch.rand(n_rays, S - 1, generator=g) - 0.5) * 0.9 * w
        ts, te = edges[:, :-1].contiguous(), edges[:, 1:].contiguous()
    bg = torch.rand(3, generator=g)
    proj = {n: torch.randn(P * n_view, Hh, Ww, c, generator=g) for n, c in KEYS}
    kn = dict(knobs, exact_f32=False, wgrad_f32=False)
    if os.environ.get("TT_FUZZ_KNOBS"):  # e.g. TT_FUZZ_KNOBS="tile_sb=0,tile_chunk=0,wgrad_f32=1"
        for kv in os.environ["TT_FUZZ_KNOBS"].split(","):
            k, v = kv.split("=")
            kn[k] = type(kn[k])(int(v))
    if os.environ.get("TT_FUZZ_RC"):  # e.g. TT_FUZZ_RC="inv_std=10"
        for kv in os.environ["TT_FUZZ_RC"].split(","):
            k, v = kv.split("=")
            rck[k] = float(v)
    _, _, gd = _hip_grads(mods, cache, sw, fw, ro, rd, ts, te, bg, cd, c2w, proj, dict(rck, **kn))
    _, _, gx = _hip_grads(mods, cache, sw, fw, ro, rd, ts, te, bg, cd, c2w, proj, dict(rck, **dict(kn, exact_f32=True)))
    _, _, g32 = _oracle_grads(torch.float32, cache, sw, fw, ro, rd, ts, te, bg, cd, c2w, proj, rck)
    _, _, g64 = _oracle_grads(torch.float64, cache, sw, fw, ro, rd, ts, te, bg, cd, c2w, proj, rck)
    verbose = len(seeds) <= 40
    if verbose:
        print(f"seed {seed}: P{P} v{n_view} R{R} {Hh}x{Ww} S{S} {rck}")
    for i, n in enumerate(names):
        if float(g64[i].abs().max()) == 0:
            continue
        d64, x64, o64 = rel(gd[i], g64[i]), rel(gx[i], g64[i]), rel(g32[i], g64[i])
        d32, x32 = rel(gd[i], g32[i]), rel(gx[i], g32[i])
        r = d64 / max(o64, 1e-30)
        if d64 > 1e-4:
            worst[(seed, n)] = (r, d64, x64, o64, d32, x32)
        if verbose:
            print(f"   {n:12s} vs fp64: default {d64:.2e} exact {x64:.2e} fp32-oracle {o64:.2e} | vs fp32 oracle: default {d32:.2e} "
                  f"exact {x32:.2e} | default/oracle error ratio {r:.2f}")
print("cases with default-vs-fp64 > 1e-4, by error ratio against the fp32 oracle's own error:")
for k, v in sorted(worst.items(), key=lambda kv: -kv[1][0])[:40]:
    print(k, "ratio %.2f default64 %.2e exact64 %.2e oracle64 %.2e default32 %.2e exact32 %.2e" % v)
