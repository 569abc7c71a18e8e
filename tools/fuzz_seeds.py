"""Dev tool: for given fuzz seeds (tests/test_gpu_fuzz.py) print every gradient's distance from the fp32 oracle in the three
precision modes, next to the oracle's own order sensitivity (|fp32 - fp32'|) and distance from fp64.  usage: python tools/fuzz_seeds.py 288 225 ...  | range:A:B"""
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
from triplaneturbo_amd import _lib  # noqa: E402
if os.environ.get("TT_USE_TUNING"):
    _lib.use_tuning_build()
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
    kn = dict(knobs)
    if os.environ.get("TT_FUZZ_KNOBS"):  # e.g. TT_FUZZ_KNOBS="tile_sb=0,tile_chunk=0"
        for kv in os.environ["TT_FUZZ_KNOBS"].split(","):
            k, v = kv.split("=")
            kn[k] = type(kn[k])(int(v)) if k in kn else bool(int(v))  # (wgrad_f32=1 / bwd_pair=1: tuning build, TT_USE_TUNING=1)
    if os.environ.get("TT_FUZZ_RC"):  # e.g. TT_FUZZ_RC="inv_std=10"
        for kv in os.environ["TT_FUZZ_RC"].split(","):
            k, v = kv.split("=")
            rck[k] = float(v)
    from parity import kink_free_rays  # noqa: E402
    if not os.environ.get("TT_FUZZ_NO_KINK_MASK"):
        keep = kink_free_rays(cache, sw, fw, ro, rd, ts, te, n_view)
        proj = {n: v * keep.view(P * n_view, Hh, Ww, 1).to(v.dtype) for n, v in proj.items()}
    gm = {}
    for mode in os.environ.get("TT_FUZZ_MODES", "split3,f32,split2").split(","):
        _, _, gm[mode] = _hip_grads(mods, cache, sw, fw, ro, rd, ts, te, bg, cd, c2w, proj, dict(rck, **dict(kn, precision=mode)))
    _, _, g32 = _oracle_grads(torch.float32, cache, sw, fw, ro, rd, ts, te, bg, cd, c2w, proj, rck)
    g32a = [_oracle_grads(torch.float32, cache, sw, fw, ro, rd, ts, te, bg, cd, c2w, proj, rck, alt_order=lv)[2] for lv in (1, 2, 3)]
    _, _, g64 = _oracle_grads(torch.float64, cache, sw, fw, ro, rd, ts, te, bg, cd, c2w, proj, rck)
    verbose = len(seeds) <= 40
    if verbose:
        print(f"seed {seed}: P{P} v{n_view} R{R} {Hh}x{Ww} S{S} {rck}")
    for i, n in enumerate(names):
        if float(g64[i].abs().max()) == 0:
            continue
        d = {m: min(rel(gm[m][i], e[i]) for e in [g32] + g32a) for m in gm}  # nearest of the fp32 evaluations
        ev = [g32] + g32a
        sens = max(rel(ev[k][i], ev[j][i]) for k in range(len(ev)) for j in range(k))
        o64 = rel(g32[i], g64[i])
        bar = max(1e-4, 1.5 * sens)
        for m in gm:
            if d[m] > 1e-4:
                worst[(seed, n, m)] = (d[m] / bar, d[m], sens, o64, rel(gm[m][i], g64[i]))
        if verbose:
            print(f"   {n:12s} vs fp32 oracle: " + " ".join(f"{m} {d[m]:.2e}" for m in d) + " | fp32 order "
                  f"sensitivity {sens:.2e}  fp32 vs fp64 {o64:.2e}")
print("gradients further than 1e-4 from the fp32 oracle (mode; ratio to the bar max(1e-4, 1.5 x order sensitivity)):")
for k, v in sorted(worst.items(), key=lambda kv: -kv[1][0]):
    print(k, "ratio_to_bar %.2f  vs_fp32 %.2e  order_sens %.2e  fp32_vs_fp64 %.2e  vs_fp64 %.2e" % v)
over = {}
for (seed, n, m), v in worst.items():
    if v[0] > 1.0:
        over.setdefault(m, set()).add(seed)
print("seeds over the bar per mode:", {m: sorted(v) for m, v in over.items()})
