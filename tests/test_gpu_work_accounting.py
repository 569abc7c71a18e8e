"""The device-side work accounting (tt_render_cfg.stats) of the three decode kernels, in every precision mode.  (The
wave-pair texture backward this file also covered in round 4 left the product library: tuning build only.)"""
import pytest
import torch

from oracle import cpu_ref as O

from parity import PRECISIONS
from test_gpu_backward import KEYS, _hip_grads, mods  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def _inputs(P, R, n_view, Hh, Ww, S, seed, near=0.3, far=3.2):
    g = torch.Generator().manual_seed(seed)
    cache = torch.randn(P, 6, 32, R, R, generator=g) * 0.5
    sw = O.init_mlp_weights([32, 64, 64, 1], g)
    fw = O.init_mlp_weights([96, 64, 64, 3], g)
    ro, rd, c2w, cd = O.make_cameras(P * n_view, Hh, Ww)
    ts, te = O.uniform_intervals(P * n_view * Hh * Ww, S, near, far)
    proj = {n: torch.randn(P * n_view, Hh, Ww, c, generator=g) for n, c in KEYS}
    return cache, sw, fw, ro, rd, ts, te, torch.ones(3), cd, c2w, proj


@pytest.mark.parametrize("precision", PRECISIONS)
def test_work_accounting_counts_what_the_kernels_execute(mods, precision):
    """tt_render_cfg.stats (rows: forward / geometry backward / texture backward; columns: tile steps visited, tile steps
    that ran the MLP chain, in-bounds (plane, sample) pairs of the gathers that ran).  A ray bundle whose samples all lie
    inside the plane cube finds every pair in bounds; one moved to (5, 5, 5) finds none, executes no tile step (exact
    skip: zeros padding makes every feature and every feature gradient zero) and returns zero plane gradients."""
    P, R, nv, Hh, Ww, S = 1, 32, 1, 8, 8, 32
    cache, sw, fw, ro, rd, ts, te, bg, cd, c2w, proj = _inputs(P, R, nv, Hh, Ww, S, 5)
    n = Hh * Ww * S
    rck = dict(inv_std=20.0, rgb_grad_shrink=1.0, cos_anneal_ratio=1.0, precision=precision)
    for inside in (True, False):
        if inside:  # camera distance 1.56: t in [1.3, 1.8] keeps |p| <= 0.81 on every ray of the 8x8 view
            ts, te = O.uniform_intervals(Hh * Ww, S, 1.3, 1.8)
            o = ro
        else:
            ts, te = O.uniform_intervals(Hh * Ww, S, 0.3, 1.0)
            o = ro + 5.0
        stats = torch.zeros((3, 4), dtype=torch.int64, device="cuda")
        _, _, grads = _hip_grads(mods, cache, sw, fw, o, rd, ts, te, bg, cd, c2w, proj, dict(rck, stats=stats))
        st = stats.cpu().tolist()
        for r, (visited, executed, inb, _) in enumerate(st):
            assert visited == n // 32, (r, st)  # every 32-sample tile step is looked at exactly once
            assert executed <= visited and inb <= 3 * 32 * executed, (r, st)
            if inside:
                assert inb == 3 * 32 * executed, (r, st)
            else:
                assert executed == 0 and inb == 0, (r, st)
        if inside:
            assert st[0][1] == st[0][0], st  # the forward runs every tile step of an in-cube bundle
            assert st[1][1] > 0 and st[2][1] > 0, st
        else:
            assert float(grads[0].abs().max()) == 0.0
