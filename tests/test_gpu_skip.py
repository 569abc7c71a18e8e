"""Opt-in approximation of the training backward (north_star: "wavefront ballot ... for ray compaction and early
termination"; SURVEY 8d: "OFF for parity, ON for throughput with the induced error reported"): with
RenderConfig.skip_eps_tex / skip_eps_geo > 0 the backward kernels skip (one ballot per tile) 32-sample tiles whose
upstream gradients are all below the threshold.  Default 0 = exact.  Here: eps = 0 is the exact path; eps > 0 changes
only the gradients of the skipped network, by an amount that is measured, reported and bounded."""
import pytest
import torch

from oracle import cpu_ref as O

from parity import report

pytestmark = pytest.mark.gpu


def _scene():
    from triplaneturbo_amd import ops
    g = torch.Generator().manual_seed(41)
    P, R, Hh, Ww, S = 1, 64, 32, 32, 96
    cache = (torch.randn(P, 6, 32, R, R, generator=g) * 0.5).cuda()
    sw = [w.cuda() for w in O.init_mlp_weights([32, 64, 64, 1], g)]
    fw = [w.cuda() for w in O.init_mlp_weights([96, 64, 64, 3], g)]
    ro, rd, c2w, cd = [t.cuda() for t in O.make_cameras(1, Hh, Ww)]
    ts, te = [t.cuda() for t in O.uniform_intervals(Hh * Ww, S, 0.1, 4.0)]
    proj = torch.randn(1, Hh, Ww, 3, generator=g).cuda()
    return ops, cache, sw, fw, ro, rd, c2w, cd, ts, te, proj


def _grads(rc, eikonal):
    from triplaneturbo_amd import functional
    ops, cache, sw, fw, ro, rd, c2w, cd, ts, te, proj = _scene()
    c = cache.clone().requires_grad_(True)
    sws = [w.clone().requires_grad_(True) for w in sw]
    fws = [w.clone().requires_grad_(True) for w in fw]
    out = functional.volume_render(c, sws, fws, ro, rd, ts, te, torch.ones(3, device="cuda"), cd, c2w, rc, training=True)
    loss = (out["comp_rgb"] * proj).sum() + (out["opacity"] ** 2 + 0.01).sqrt().mean()
    if eikonal:
        loss = loss + ops.eikonal_loss(out["sdf_grad"])
    g = torch.autograd.grad(loss, [c] + sws + fws)
    # the per-sample upstream colour gradient the texture kernel thresholds: cbar = w g_rgb 1.002 s (1 - s)
    s = torch.sigmoid(out["features"].detach())
    grgb = proj.reshape(-1, 3).repeat_interleave(ts.shape[1], dim=0)
    cb = (out["weights"].detach() * grgb * 1.002 * s * (1 - s)).abs().sum(-1)
    return [t.double() for t in g], cb


def _rel(a, b):
    return ((a - b).norm() / b.norm().clamp_min(1e-300)).item()


def test_skip_thresholds_trade_a_bounded_gradient_error_for_skipped_tiles():
    from triplaneturbo_amd import ops
    exact, cb = _grads(ops.RenderConfig(), eikonal=False)
    again, _ = _grads(ops.RenderConfig(skip_eps_tex=0.0, skip_eps_geo=0.0), eikonal=False)
    for a, b in zip(again, exact):
        assert _rel(a, b) < 2e-5  # eps = 0 IS the exact path (float-atomic summation order only)
    rows = {}
    for frac in (1e-5, 1e-4, 1e-3):
        eps = frac * cb.max().item()
        got, _ = _grads(ops.RenderConfig(skip_eps_tex=eps), eikonal=False)
        below = (cb <= eps).double().mean().item()
        # geometry side untouched
        assert _rel(got[0][:, :3], exact[0][:, :3]) < 2e-5 and all(_rel(got[k], exact[k]) < 2e-5 for k in (1, 2, 3))
        err = {"planes_tex": _rel(got[0][:, 3:], exact[0][:, 3:]), "v1": _rel(got[4], exact[4]), "v2": _rel(got[5], exact[5]),
               "v3": _rel(got[6], exact[6])}
        rows[f"skip_eps_tex = {frac:g} max|cbar|"] = dict(err, samples_below_eps=below)
        # dropping everything below eps removes at most the l1 mass of the dropped upstream: error <= that fraction
        dropped_mass = (cb[cb <= eps].sum() / cb.sum()).item()
        assert max(err.values()) <= max(20 * dropped_mass, 1e-6), (frac, err, dropped_mass)
    report("opt-in backward skip (texture), 32x32 rays x 96 samples", rows)
    # geometry threshold: dense under the eikonal term (nothing to skip, result unchanged) ...
    ex_e, _ = _grads(ops.RenderConfig(), eikonal=True)
    got_e, _ = _grads(ops.RenderConfig(skip_eps_geo=1e-12), eikonal=True)
    assert all(_rel(a, b) < 2e-5 for a, b in zip(got_e, ex_e))
    # ... and with a huge threshold every tile is skipped: geometry gradients vanish, texture gradients are untouched
    got_h, _ = _grads(ops.RenderConfig(skip_eps_geo=1e30), eikonal=False)
    assert got_h[0][:, :3].abs().max().item() == 0.0 and all(got_h[k].abs().max().item() == 0.0 for k in (1, 2, 3))
    assert _rel(got_h[0][:, 3:], exact[0][:, 3:]) < 2e-5
