"""The ray march alone (tt_march_fwd / tt_march_bwd, the nerfacc boundary: NeuS alpha -> render_weight_from_alpha ->
accumulate_along_rays x5) against the oracle on given per-sample sdf / sdf_grad / features.  S covers every kernel
variant: all passes of a ray held in registers (S <= 64, <= 128, <= 256) and the streaming form (S > 256), with
ragged last passes.  Backward = autograd of the oracle in fp64 (tolerance: as close as the fp32 oracle, x4)."""
import pytest
import torch
import torch.nn.functional as F

from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu


def _inputs(n_rays, S, seed, dtype):
    g = torch.Generator().manual_seed(seed)
    rd = F.normalize(torch.randn(n_rays, 3, generator=g), dim=-1)
    edges = torch.sort(torch.rand(n_rays, S + 1, generator=g) * 3.5 + 0.1, dim=1).values
    ts, te = edges[:, :-1].contiguous(), edges[:, 1:].contiguous()
    sdf = torch.randn(n_rays * S, 1, generator=g) * 0.05
    sdf_grad = torch.randn(n_rays * S, 3, generator=g)
    sdf_grad[::17] = 0.0  # F.normalize eps branch
    feat = torch.randn(n_rays * S, 3, generator=g) * 2
    return [t.to(dtype) for t in (rd, ts, te, sdf, sdf_grad, feat)]


def _oracle_march(rd, ts, te, sdf, sdf_grad, feat, inv_std, ratio):
    n_rays, S = ts.shape
    tm = ((ts + te) / 2.0).reshape(-1, 1)
    dt = (te - ts).reshape(-1, 1)
    ridx = torch.arange(n_rays).unsqueeze(-1).expand(-1, S).reshape(-1)
    normal = F.normalize(sdf_grad, dim=-1)
    alpha = O.get_alpha(sdf, normal, rd[ridx], dt, inv_std, ratio)
    w2, tr2 = O.render_weight_from_alpha(alpha.reshape(n_rays, S))
    w = w2.reshape(-1, 1)
    acc = lambda v: (w if v is None else w * v).reshape(n_rays, S, -1).sum(dim=1)
    depth = acc(tm)
    return {"opacity": acc(None), "depth": depth, "rgb_fg": acc(O.sigmoid_mipnerf(feat)),
            "z_variance": acc((tm - depth[ridx]) ** 2), "normal_acc": acc(normal), "weights": w,
            "trans": tr2.reshape(-1, 1)}


@pytest.mark.parametrize("S", [7, 64, 65, 128, 193, 256, 300])
def test_march_forward_and_backward_match_oracle(S):
    from triplaneturbo_amd import ops
    n_rays, inv_std, ratio = 37, 40.0, 0.3
    rc = ops.RenderConfig(inv_std=inv_std, cos_anneal_ratio=ratio)
    x32 = _inputs(n_rays, S, 100 + S, torch.float32)
    dev = [t.cuda() for t in x32]
    out = ops.march_forward_raw(dev[0], dev[1], dev[2], dev[3], dev[4], dev[5], rc)

    ref = {}
    for dt in (torch.float32, torch.float64):
        x = [t.to(dt) for t in x32]
        for t in x[3:]:
            t.requires_grad_(True)
        o = _oracle_march(*x, inv_std, ratio)
        g = torch.Generator().manual_seed(5)
        ups = {k: torch.randn(v.shape, generator=g).to(dt) for k, v in o.items() if k != "trans"}
        loss = sum((o[k] * ups[k]).sum() for k in ups)
        grads = torch.autograd.grad(loss, x[3:5])
        ref[dt] = (o, ups, grads)
    o32, _, g32 = ref[torch.float32]
    o64, ups, g64 = ref[torch.float64]
    for k in o64:
        e_hip = (out[k].cpu().double() - o64[k].detach()).abs().max().item()
        e_cpu = (o32[k].detach().double() - o64[k].detach()).abs().max().item()
        assert e_hip <= max(4 * e_cpu, 2e-6), (k, S, e_hip, e_cpu)

    u = {k: v.float().cuda() for k, v in ups.items()}
    ws = ops.march_backward_raw(dev[0], dev[1], dev[2], out, dev[3], dev[4], dev[5], rc, g_opacity=u["opacity"],
                                g_depth=u["depth"], g_rgb_fg=u["rgb_fg"], g_z_variance=u["z_variance"],
                                g_normal_acc=u["normal_acc"], g_weights=u["weights"])
    got = ws.cpu().double()
    want = torch.cat([g64[0], g64[1]], dim=1)
    want32 = torch.cat([g32[0], g32[1]], dim=1).double()
    live = torch.ones(n_rays * S, dtype=torch.bool)
    live[::17] = False  # zero sdf_grad: d normalize / d g = 1/eps there (1e12), compare separately in relative terms
    rel = lambda a, b: ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
    assert rel(got[live], want[live]) <= max(1e-4, 3 * rel(want32[live], want[live])), S
    assert rel(got[~live], want[~live]) <= max(1e-4, 3 * rel(want32[~live], want[~live])), S


def test_march_equals_the_fused_forward():
    """tt_render_fwd = decode kernel + the same march kernel: feeding its per-sample outputs back through tt_march_fwd
    reproduces its per-ray outputs bit for bit."""
    from triplaneturbo_amd import ops
    g = torch.Generator().manual_seed(3)
    P, R, Hh, Ww, S = 1, 32, 6, 5, 70
    cache = (torch.randn(P, 6, 32, R, R, generator=g) * 0.5).cuda()
    sw = [w.cuda() for w in O.init_mlp_weights([32, 64, 64, 1], g)]
    fw = [w.cuda() for w in O.init_mlp_weights([96, 64, 64, 3], g)]
    ro, rd, _, _ = O.make_cameras(1, Hh, Ww)
    ro, rd = ro.reshape(-1, 3).cuda(), rd.reshape(-1, 3).cuda()
    ts, te = O.uniform_intervals(Hh * Ww, S, 0.3, 3.2)
    ts, te = ts.cuda(), te.cuda()
    rc = ops.RenderConfig(inv_std=50.0)
    full = ops.render_forward_raw(ops.planes_pack(cache), sw, fw, ro, rd, ts, te, Hh * Ww, rc, image_w=Ww)
    again = ops.march_forward_raw(rd, ts, te, full["sdf"], full["sdf_grad"], full["features"], rc)
    for k in again:
        assert torch.equal(again[k], full[k]), k
