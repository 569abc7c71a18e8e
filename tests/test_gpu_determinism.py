"""Run-to-run determinism of the atomics-free kernels.  Regression test for a nondeterministic corruption found in round
2: with the textbook in-bounds logic (lane masks combined on the scalar ALU, then v_cndmask) the bilinear weight w[2] of
lanes 48..63 was occasionally stale when a SIMD ran a single wave -- ~5 of 9375 tiles of a 300 k-point query per launch,
different tiles every launch, invisible to tolerance-based parity tests on coherent rays.  corners_setup
(csrc/tt_device.h) now avoids mask logic; identical launches must give bit-identical results, and they must match the
oracle on every point of a sample of tiles."""
import pytest
import torch

from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu


def _setup(seed=8, n_v=300_000):
    dev = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(seed)
    cache = torch.randn(1, 6, 32, 256, 256, generator=gen) * 0.5
    sw = O.init_mlp_weights([32, 64, 64, 1], gen)
    fw = O.init_mlp_weights([96, 64, 64, 3], gen)
    # incoherent points (mesh-vertex colouring, few_step...:402-430): every lane of a tile in a different texel
    v = torch.nn.functional.normalize(torch.randn(n_v, 3, generator=gen), dim=-1) * (
        0.5 + 0.05 * torch.randn(n_v, 1, generator=gen))
    return dev, cache, sw, fw, v


@pytest.mark.parametrize("need_normal", [False, True])
def test_point_query_is_bitwise_reproducible(need_normal):
    from triplaneturbo_amd import ops
    dev, cache, sw, fw, v = _setup()
    packed = ops.planes_pack(cache.to(dev))
    swd, fwd, vd = [w.to(dev) for w in sw], [w.to(dev) for w in fw], v.to(dev)[None]
    ref = None
    for it in range(12):
        out = ops.query_points(packed, swd, fwd, vd, need_normal=need_normal, need_features=True)
        torch.cuda.synchronize()
        cur = [t.clone() for t in out if t is not None]
        if ref is None:
            ref = cur
            continue
        for a, b in zip(ref, cur):
            bad = (a != b).any(dim=-1).nonzero().flatten()
            assert bad.numel() == 0, (it, bad.numel(), sorted(set((bad // 32).tolist()))[:8],
                                      sorted(set((bad % 32).tolist())))
    # and the reproducible answer is the right one: whole tiles (all 32 lanes) from the tail of the launch
    tiles = torch.arange(9375 - 48, 9375, 3)
    idx = (tiles[:, None] * 32 + torch.arange(32)[None]).flatten()
    want = O.geometry_forward(v[idx][None], cache, sw, fw, output_normal=False)
    got_sdf, got_feat = ref[0][idx.to(dev)].cpu(), ref[-1][idx.to(dev)].cpu()
    assert (got_sdf - want["sdf"].reshape(-1, 1)).abs().max().item() < 2e-5
    assert (got_feat - want["features"].reshape(-1, 3)).abs().max().item() < 2e-5


def test_eval_render_is_bitwise_reproducible():
    from triplaneturbo_amd import ops
    dev, cache, sw, fw, _ = _setup(seed=9, n_v=64)
    packed = ops.planes_pack(cache.to(dev))
    swd, fwd = [w.to(dev) for w in sw], [w.to(dev) for w in fw]
    Hh = Ww = 64
    ro, rd, c2w, cd = [t.to(dev) for t in O.make_cameras(1, Hh, Ww)]
    ts, te = [t.to(dev) for t in O.uniform_intervals(Hh * Ww, 96, 0.3, 3.2)]
    ref = None
    for it in range(8):
        out = ops.render_forward_raw(packed, swd, fwd, ro.reshape(-1, 3), rd.reshape(-1, 3), ts, te, Hh * Ww,
                                     ops.RenderConfig(), image_w=Ww)
        torch.cuda.synchronize()
        cur = {k: out[k].clone() for k in ("opacity", "rgb_fg", "sdf", "sdf_grad", "features", "weights")}
        if ref is None:
            ref = cur
            continue
        for k in ref:
            assert torch.equal(ref[k], cur[k]), (it, k)


@pytest.mark.parametrize("S", [64, 61])  # 61: ragged last chunk (lanes past the end of the ray in the last tile step)
def test_training_step_gradients_spread_only_by_summation_order(S):
    """Plane and weight gradients are float-atomic sums: launches differ by summation order (~1e-6 of the largest
    element), never by a dropped or corrupted sample (which moves single texels by >= 1e-3 of it)."""
    from triplaneturbo_amd import functional, ops
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(31)
    P, R, Hh, Ww = 1, 128, 64, 64
    cache = (torch.randn(P, 6, 32, R, R, generator=g) * 0.5).to(dev).requires_grad_(True)
    sw = [w.to(dev).requires_grad_(True) for w in O.init_mlp_weights([32, 64, 64, 1], g)]
    fw = [w.to(dev).requires_grad_(True) for w in O.init_mlp_weights([96, 64, 64, 3], g)]
    ro, rd, c2w, cd = [t.to(dev) for t in O.make_cameras(1, Hh, Ww)]
    ts, te = [t.to(dev) for t in O.uniform_intervals(Hh * Ww, S, 0.3, 3.2)]
    bg = torch.ones(3, device=dev)
    rc = ops.RenderConfig(inv_std=50.0)
    params = [cache] + sw + fw

    def step():
        out = functional.volume_render(cache, sw, fw, ro, rd, ts, te, bg, cd, c2w, rc, training=True)
        loss = out["comp_rgb"].mean() + out["comp_normal_cam_vis"].mean() + out["opacity"].mean() + \
            ((out["sdf_grad"].norm(dim=-1) - 1) ** 2).mean()
        grads = torch.autograd.grad(loss, params)
        torch.cuda.synchronize()
        return [t.clone() for t in grads], out["comp_rgb"].detach().clone()

    ref, ref_rgb = step()
    for it in range(10):
        cur, rgb = step()
        assert torch.equal(rgb, ref_rgb), it  # the forward has no atomics: bit-identical
        for a, b in zip(ref, cur):
            assert ((a - b).abs().max() / a.abs().max()).item() < 2e-5, it


def test_eval_render_and_field_query_are_bitwise_reproducible():
    """The fused eval renderer (ballot-based early termination) and the implicit-field query have no atomics either."""
    from triplaneturbo_amd import ops
    dev, cache, sw, fw, _ = _setup(seed=10, n_v=64)
    packed = ops.planes_pack(cache.to(dev))
    swd, fwd = [w.to(dev) for w in sw], [w.to(dev) for w in fw]
    gen = torch.Generator().manual_seed(3)
    dw = [w.to(dev) for w in O.init_mlp_weights([32, 64, 64, 3], gen)]
    Hh = Ww = 96
    ro, rd, c2w, cd = [t.to(dev) for t in O.make_cameras(1, Hh, Ww)]
    ts, te = [t.to(dev) for t in O.uniform_intervals(Hh * Ww, 64, 0.3, 3.2)]
    pts = (torch.rand(1, 200_000, 3, generator=gen) * 2 - 1).to(dev)
    ref = None
    for it in range(8):
        ev = ops.render_eval_raw(packed, swd, fwd, ro.reshape(-1, 3), rd.reshape(-1, 3), ts, te, Hh * Ww,
                                 ops.RenderConfig(), image_w=Ww, transmittance_eps=1e-4, weight_eps=1e-6)
        sdf, deform = ops.query_field(packed, swd, dw, pts)
        torch.cuda.synchronize()
        cur = {k: ev[k].clone() for k in ("opacity", "depth", "rgb_fg", "normal_acc")}
        cur["field_sdf"], cur["field_deform"] = sdf.clone(), deform.clone()
        if ref is None:
            ref = cur
            continue
        for k in ref:
            assert torch.equal(ref[k], cur[k]), (it, k)


def test_gradient_wrt_query_points_is_bitwise_reproducible():
    """tt_points_bwd_x (d/d query points, incl. the second-order cross derivative) accumulates nothing across lanes: twelve
    launches on 100 k incoherent points must agree bit for bit (a stale lane mask would show up as per-tile differences,
    as it did in the per-point forward: tools/stress_export.py)."""
    import triplaneturbo_amd as tt
    dev = torch.device("cuda", 0)
    torch.manual_seed(31)
    g = tt.find("few-step-triplane-dual-stable-diffusion")({}).to(dev)
    for w_ in g.parameters():
        w_.requires_grad_(False)  # points only: no atomics anywhere in the backward
    gen = torch.Generator().manual_seed(32)
    cache = (torch.randn(1, 6, 32, 128, 128, generator=gen) * 0.5).to(dev)
    pts = (torch.rand(1, 100_000, 3, generator=gen) * 2.1 - 1.05).to(dev)
    proj = {k: torch.randn(100_000, c, generator=gen).to(dev) for k, c in (("sdf", 1), ("features", 3), ("sdf_grad", 3))}
    ref = None
    for it in range(12):
        x = pts.clone().requires_grad_(True)
        out = g(x, cache, output_normal=True)
        gx, = torch.autograd.grad(sum((out[k] * proj[k]).sum() for k in proj), [x])
        torch.cuda.synchronize()
        if ref is None:
            ref = gx.clone()
            assert torch.isfinite(ref).all()
        else:
            assert torch.equal(ref, gx), (it, (ref != gx).any(-1).nonzero().flatten()[:8].tolist())


def test_ray_march_forward_and_backward_are_bitwise_reproducible():
    """tt_march_fwd / tt_march_bwd have no atomics (one wave per ray, DPP scans): every output -- including the
    (n_rays*S, 4) upstream workspace the geometry backward consumes -- must be bit-identical over repeated launches."""
    from triplaneturbo_amd import ops
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(41)
    n_rays, S = 3001, 193
    rd = torch.nn.functional.normalize(torch.randn(n_rays, 3, generator=g), dim=-1).to(dev)
    ts, te = [t.to(dev) for t in O.uniform_intervals(n_rays, S, 0.1, 4.0)]
    sdf = (torch.randn(n_rays * S, 1, generator=g) * 0.05).to(dev)
    sdf_grad = torch.randn(n_rays * S, 3, generator=g).to(dev)
    feat = torch.randn(n_rays * S, 3, generator=g).to(dev)
    ups = dict(g_opacity=torch.randn(n_rays, 1, generator=g).to(dev), g_depth=torch.randn(n_rays, 1, generator=g).to(dev),
               g_rgb_fg=torch.randn(n_rays, 3, generator=g).to(dev), g_z_variance=torch.randn(n_rays, 1, generator=g).to(dev),
               g_normal_acc=torch.randn(n_rays, 3, generator=g).to(dev),
               g_sdf_grad=(torch.randn(n_rays * S, 3, generator=g) * 1e-3).to(dev))
    rc = ops.RenderConfig(inv_std=60.0, cos_anneal_ratio=0.5)
    fwd0 = ops.march_forward_raw(rd, ts, te, sdf, sdf_grad, feat, rc)
    ws0 = ops.march_backward_raw(rd, ts, te, fwd0, sdf, sdf_grad, feat, rc, **ups)
    assert torch.isfinite(ws0).all()
    for it in range(12):
        fwd = ops.march_forward_raw(rd, ts, te, sdf, sdf_grad, feat, rc)
        for k in fwd0:
            assert torch.equal(fwd[k], fwd0[k]), (it, k)
        ws = ops.march_backward_raw(rd, ts, te, fwd, sdf, sdf_grad, feat, rc, **ups)
        assert torch.equal(ws, ws0), it
