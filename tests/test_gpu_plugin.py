"""GPU tests through the drop-in plugin API (registry names / Config / forward signature / output keys of the
reference): eval + training renders, the importance sampler, PatchRenderer, geometry queries."""
import pytest
import torch

import triplaneturbo_amd as tt
from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda", 0)


def _build(dev, seed=0, n_samples=16, n_imp=32):
    torch.manual_seed(seed)
    g = tt.find("few-step-triplane-dual-stable-diffusion")({}).to(dev)
    m = tt.find("no-material")({})
    b = tt.find("solid-color-background")({})
    cfg = dict(estimator="importance", trainable_variance=False, learned_variance_init=0.4605,
               num_samples_per_ray=n_samples, num_samples_per_ray_importance=n_imp, near_plane=0.1, far_plane=4.0,
               rgb_grad_shrink=[0, 1, 0.01, 20000])
    r = tt.find("generative-space-sdf-volume-renderer")(cfg, geometry=g, material=m, background=b).to(dev)
    return g, m, b, r, cfg


def _weights(g):
    sw, fw = g.mlp_weights()
    return [w.detach().cpu() for w in sw], [w.detach().cpu() for w in fw]


def test_eval_render_matches_oracle_and_keys(dev):
    g, m, b, r, cfg = _build(dev)
    r.eval()
    assert r.eval_termination_eps == 0.0  # the DEFAULT marches every sample like the reference (opt-in: test_gpu_eval)
    gen = torch.Generator().manual_seed(1)
    P, n_view, Hh, Ww = 1, 2, 6, 8
    cache = torch.randn(P, 6, 32, 32, 32, generator=gen) * 0.5
    ro, rd, c2w, cd = O.make_cameras(P * n_view, Hh, Ww)
    ts, te = O.uniform_intervals(P * n_view * Hh * Ww, 24, 0.3, 3.2)
    with torch.no_grad():
        out = r(ro.to(dev), rd.to(dev), None, None, space_cache=cache.to(dev), text_embed=torch.zeros(P, 77, 1024),
                camera_distances=cd.to(dev), c2w=c2w.to(dev), t_starts=ts.to(dev), t_ends=te.to(dev))
    # eval-mode key set of the reference (renderer :442-449,460,477,503-504; no training extras)
    assert set(out) == {"comp_rgb", "comp_rgb_fg", "comp_rgb_bg", "opacity", "depth", "z_variance", "disparity",
                        "comp_normal", "comp_normal_cam_vis", "comp_normal_cam_vis_white"}
    sw, fw = _weights(g)
    want = O.render(cache.double(), [w.double() for w in sw], [w.double() for w in fw], ro.double(), rd.double(),
                    ts.double(), te.double(), torch.ones(3).double(), cd.double(), c2w.double(), create_graph=False,
                    training=False, inv_std=float(r.variance.inv_std))
    w32 = O.render(cache, sw, fw, ro, rd, ts, te, torch.ones(3), cd, c2w, create_graph=False, training=False,
                   inv_std=float(r.variance.inv_std))
    for k in ("comp_rgb", "opacity", "depth", "z_variance", "disparity", "comp_normal_cam_vis"):
        e_hip = (out[k].cpu().double() - want[k]).abs().max().item()
        e_cpu = (w32[k].double() - want[k]).abs().max().item()
        assert e_hip <= max(4 * e_cpu, 2e-5), (k, e_hip, e_cpu)


def test_importance_sampler_matches_oracle_contract(dev):
    g, m, b, r, cfg = _build(dev, n_samples=16, n_imp=32)
    r.eval()  # stratified = False
    gen = torch.Generator().manual_seed(2)
    cache = torch.randn(1, 6, 32, 32, 32, generator=gen) * 0.5
    ro, rd, c2w, cd = O.make_cameras(1, 5, 6)
    ts, te = r.sample(cache.to(dev), ro.to(dev), rd.to(dev))
    assert ts.shape == (30, 32 + 16 + 1)
    sw, fw = _weights(g)

    def sdf_fn(a, b2):  # oracle geometry at the interval mid-points
        pos = ro.reshape(-1, 1, 3) + rd.reshape(-1, 1, 3) * ((a + b2) / 2)[..., None]
        o = O.geometry_forward(pos.reshape(1, -1, 3), cache, sw, fw, output_normal=False)
        return o["sdf"].reshape(a.shape)

    wts, wte = O.importance_sampling(sdf_fn, 30, 32, 16, 0.1, 4.0, float(r.variance.inv_std), r.render_step_size)
    # same contract, fp32 sdf differences of ~1e-6 move the resampled edges by <1e-4
    assert (ts.cpu() - wts).abs().max().item() < 2e-4 and (te.cpu() - wte).abs().max().item() < 2e-4
    assert (ts[:, 1:] >= ts[:, :-1]).all()


def test_training_render_through_plugin_backward(dev):
    g, m, b, r, cfg = _build(dev)
    r.train()
    r.update_step(0, 10000)  # rgb_grad_shrink = 0.505
    gen = torch.Generator().manual_seed(3)
    P, n_view, Hh, Ww = 2, 2, 4, 8
    cache = (torch.randn(P, 6, 32, 32, 32, generator=gen) * 0.5).to(dev).requires_grad_(True)
    ro, rd, c2w, cd = O.make_cameras(P * n_view, Hh, Ww)
    out = r(ro.to(dev), rd.to(dev), None, torch.ones(3, device=dev), space_cache=cache,
            text_embed=torch.zeros(P, 77, 1024), camera_distances=cd.to(dev), c2w=c2w.to(dev))
    for k in ("weights", "t_points", "t_intervals", "t_dirs", "ray_indices", "points", "sdf", "sdf_orig", "features",
              "normal", "shading_normal", "sdf_grad", "inv_std"):  # renderer :532-545
        assert k in out, k
    S = cfg["num_samples_per_ray"] + cfg["num_samples_per_ray_importance"] + 1
    assert out["weights"].shape == (P * n_view * Hh * Ww * S, 1)
    loss = out["comp_rgb"].mean() + (out["opacity"] ** 2 + 0.01).sqrt().mean() + \
        ((out["sdf_grad"].norm(dim=-1) - 1) ** 2).mean()
    loss.backward()
    assert cache.grad is not None and torch.isfinite(cache.grad).all() and cache.grad.abs().sum() > 0
    for w in g.parameters():
        assert w.grad is not None and torch.isfinite(w.grad).all()


def test_trainable_variance_gradient_matches_oracle(dev):
    """trainable_variance=True (the reference class default, generative_space_sdf_volume_renderer.py:53,82;
    neus_volume_renderer.py:26-37): inv_std = exp(10 p) lives on the device, the kernels read it there, and
    d loss / d p comes back through tt_render_bwd_geo's per-ray partials.  Checked against autograd through the oracle
    (fp64 and fp32) with p as a leaf; every other gradient must equal the frozen-variance render's."""
    torch.manual_seed(0)
    g = tt.find("few-step-triplane-dual-stable-diffusion")({}).to(dev)
    m, b = tt.find("no-material")({}), tt.find("solid-color-background")({})
    base = dict(estimator="importance", learned_variance_init=0.35, num_samples_per_ray=16,
                num_samples_per_ray_importance=32, near_plane=0.1, far_plane=4.0)
    rt = tt.find("generative-space-sdf-volume-renderer")(dict(base, trainable_variance=True), geometry=g, material=m,
                                                         background=b).to(dev)
    rf = tt.find("generative-space-sdf-volume-renderer")(dict(base, trainable_variance=False), geometry=g, material=m,
                                                         background=b).to(dev)
    rt.train(), rf.train()
    gen = torch.Generator().manual_seed(5)
    P, n_view, Hh, Ww, S = 1, 2, 5, 7, 40
    cache0 = torch.randn(P, 6, 32, 32, 32, generator=gen) * 0.5
    ro, rd, c2w, cd = O.make_cameras(P * n_view, Hh, Ww)
    ts, te = O.uniform_intervals(P * n_view * Hh * Ww, S, 0.3, 3.2)
    proj = {k: torch.randn(P * n_view, Hh, Ww, c, generator=gen) for k, c in (("comp_rgb", 3), ("opacity", 1), ("depth", 1))}

    def run_hip(r):
        cache = cache0.to(dev).requires_grad_(True)
        for p_ in list(g.parameters()) + list(r.parameters()):
            p_.grad = None
        out = r(ro.to(dev), rd.to(dev), None, torch.ones(3, device=dev), space_cache=cache,
                text_embed=torch.zeros(P, 77, 1024), camera_distances=cd.to(dev), c2w=c2w.to(dev), t_starts=ts.to(dev),
                t_ends=te.to(dev))
        loss = O.synthetic_loss(out, {k: v.to(dev) for k, v in proj.items()})
        loss.backward()
        return out, loss.item(), cache.grad.detach().cpu().double(), [w.grad.detach().cpu().double() for w in g.parameters()]

    out_t, l_t, gc_t, gw_t = run_hip(rt)
    gp = rt.variance._inv_std.grad.item()
    out_f, l_f, gc_f, gw_f = run_hip(rf)
    assert rf.variance._inv_std.grad is None
    # same forward, same gradients of everything else (the device scalar is torch's fp32 exp(10 p), the host value the
    # rounded double exp: they may differ in the last bit)
    assert abs(l_t - l_f) <= 2e-5 * abs(l_f)
    assert (gc_t - gc_f).norm() <= 1e-4 * gc_f.norm()
    for a_, b_ in zip(gw_t, gw_f):
        assert (a_ - b_).norm() <= 1e-4 * b_.norm()
    assert out_t["inv_std"].requires_grad and abs(out_t["inv_std"].item() - float(rt.variance.inv_std)) < 1e-3

    sw, fw = _weights(g)

    def run_oracle(dtype):
        p = torch.tensor(0.35, dtype=dtype, requires_grad=True)
        inv_std = torch.exp(p * 10.0).clamp(1.0e-6, 1.0e6)
        out = O.render(cache0.to(dtype), [w.to(dtype) for w in sw], [w.to(dtype) for w in fw], ro.to(dtype), rd.to(dtype),
                       ts.to(dtype), te.to(dtype), torch.ones(3, dtype=dtype), cd.to(dtype), c2w.to(dtype), inv_std=inv_std)
        loss = O.synthetic_loss(out, {k: v.to(dtype) for k, v in proj.items()})
        return torch.autograd.grad(loss, [p])[0].item()

    g64, g32 = run_oracle(torch.float64), run_oracle(torch.float32)
    assert abs(gp - g32) <= 1e-4 * abs(g32) + 1e-7, (gp, g32, g64)      # north_star's rtol against the fp32 reference math
    assert abs(gp - g64) <= max(3 * abs(g32 - g64), 1e-4 * abs(g64)), (gp, g32, g64)
    # and the sampler reads the same device scalar: identical intervals from both renderers
    a, _ = rt.sample(cache0.to(dev), ro.to(dev), rd.to(dev), generator=torch.Generator(device=dev).manual_seed(1))
    b2, _ = rf.sample(cache0.to(dev), ro.to(dev), rd.to(dev), generator=torch.Generator(device=dev).manual_seed(1))
    assert torch.allclose(a, b2, rtol=0, atol=1e-5)


def test_patch_renderer_training_and_eval(dev):
    g, m, b, r, cfg = _build(dev)
    p = tt.find("patch-renderer")({"patch_size": 8, "global_downsample": 3,
                                   "base_renderer_type": "generative-space-sdf-volume-renderer",
                                   "base_renderer": cfg}, geometry=g, material=m, background=b).to(dev)
    gen = torch.Generator().manual_seed(4)
    cache = (torch.randn(1, 6, 32, 32, 32, generator=gen) * 0.5).to(dev).requires_grad_(True)
    ro, rd, c2w, cd = O.make_cameras(2, 24, 24)
    kw = dict(space_cache=cache, text_embed=torch.zeros(1, 77, 1024), camera_distances=cd.to(dev), c2w=c2w.to(dev))
    p.train()
    torch.manual_seed(0)
    out = p(ro.to(dev), rd.to(dev), None, torch.ones(3, device=dev), **kw)
    assert out["comp_rgb"].shape == (2, 24, 24, 3) and out["opacity"].shape == (2, 24, 24, 1)
    out["comp_rgb"].sum().backward()
    assert torch.isfinite(cache.grad).all() and cache.grad.abs().sum() > 0
    # values: patch region = a direct render of the patch rays, the rest = the bilinearly upsampled global render
    # (patch_renderer.py:49-89); sampling made deterministic so the three renders see the same intervals
    p.base_renderer.randomized = False
    torch.manual_seed(5)
    with torch.no_grad():
        out = p(ro.to(dev), rd.to(dev), None, torch.ones(3, device=dev), **kw)
        torch.manual_seed(5)
        px = torch.randint(0, 24 - 8, (1,)).item()
        py = torch.randint(0, 24 - 8, (1,)).item()
        direct = p.base_renderer(ro[:, py:py + 8, px:px + 8].contiguous().to(dev),
                                 rd[:, py:py + 8, px:px + 8].contiguous().to(dev), None, torch.ones(3, device=dev), **kw)
        F = torch.nn.functional
        lo_o = F.interpolate(ro.permute(0, 3, 1, 2), (8, 8), mode="bilinear").permute(0, 2, 3, 1).contiguous()
        lo_d = F.interpolate(rd.permute(0, 3, 1, 2), (8, 8), mode="bilinear").permute(0, 2, 3, 1).contiguous()
        glob = p.base_renderer(lo_o.to(dev), lo_d.to(dev), None, torch.ones(3, device=dev), **kw)
    for k in ("comp_rgb", "opacity", "depth", "comp_normal"):
        torch.testing.assert_close(out[k][:, py:py + 8, px:px + 8], direct[k], rtol=1e-5, atol=1e-6)
        up = F.interpolate(glob[k].permute(0, 3, 1, 2), (24, 24), mode="bilinear").permute(0, 2, 3, 1)
        mask = torch.ones(24, 24, dtype=torch.bool, device=dev)
        mask[py:py + 8, px:px + 8] = False
        torch.testing.assert_close(out[k][:, mask], up[:, mask], rtol=1e-5, atol=1e-6)
    p.eval()
    with torch.no_grad():
        out = p(ro.to(dev), rd.to(dev), None, torch.ones(3, device=dev), **kw)
    assert out["comp_rgb"].shape == (2, 24, 24, 3)


def test_geometry_queries_match_oracle(dev):
    g, *_ = _build(dev)
    gen = torch.Generator().manual_seed(5)
    cache = torch.randn(2, 6, 32, 16, 16, generator=gen) * 0.5
    pts = torch.rand(4, 100, 3, generator=gen) * 2.4 - 1.2  # 2 views per prompt
    sw, fw = _weights(g)
    want = O.geometry_forward(pts, cache.repeat_interleave(2, 0), sw, fw, output_normal=True)
    out = g(pts.to(dev), cache.to(dev), output_normal=True)
    for k in ("sdf", "sdf_orig", "features", "sdf_grad", "normal"):
        torch.testing.assert_close(out[k].cpu(), want[k], rtol=2e-4, atol=2e-5)
    sdf = g.forward_sdf(pts.to(dev), cache.to(dev))
    torch.testing.assert_close(sdf.cpu().reshape(-1, 1), want["sdf"], rtol=2e-4, atol=2e-5)
    f, d = g.forward_field(pts.to(dev), cache.to(dev))
    assert d is None and f.shape == (4, 100, 1)
    ex = g.export(pts[:1].to(dev), cache[:1].to(dev))
    torch.testing.assert_close(ex["features"].cpu().reshape(-1, 3), want["features"][:100], rtol=2e-4, atol=2e-5)


def test_forward_field_with_deformation_head(dev):
    """the mesh renderer / exporter grid query (sdf + deformation_network, few_step...:375-394)."""
    torch.manual_seed(7)
    g = tt.find("few-step-triplane-dual-stable-diffusion")({"isosurface_deformable_grid": True}).to(dev)
    gen = torch.Generator().manual_seed(8)
    cache = torch.randn(1, 6, 32, 32, 32, generator=gen) * 0.5
    # a 12^3 grid like the isosurface helper's vertices, slightly beyond the box
    lin = torch.linspace(-1.1, 1.1, 12)
    pts = torch.stack(torch.meshgrid(lin, lin, lin, indexing="ij"), -1).reshape(1, -1, 3)
    sdf, deform = g.forward_field(pts.to(dev), cache.to(dev))
    assert sdf.shape == (1, 1728, 1) and deform.shape == (1, 1728, 3)
    sw, fw = _weights(g)
    dw = [w.detach().cpu() for w in g.deformation_network.weights()]
    want = O.geometry_forward(pts, cache, sw, fw, output_normal=False)
    torch.testing.assert_close(sdf.cpu().reshape(-1, 1), want["sdf"], rtol=2e-4, atol=2e-5)
    torch.testing.assert_close(deform.cpu().reshape(-1, 3), O.vanilla_mlp(want["enc_geo"], dw), rtol=2e-4, atol=2e-5)
    assert list(g.state_dict().keys())[-3:] == [f"deformation_network.layers.{i}.weight" for i in (0, 2, 4)]


def test_patch_renderer_composite_equals_reference_vectors(dev, golden_dir):
    """Our PatchRenderer (one HIP kernel per key: tt_patch_composite_fwd) against the REFERENCE's PatchRenderer.forward
    run on the same deterministic base renderer with the same seed for the patch position
    (tests/golden/make_golden_renderer.py -> reference_renderer.npz), and the kernel's backward against autograd of the
    reference's torch ops (F.interpolate + slice assignment, patch_renderer.py:74-88)."""
    import os

    import numpy as np
    from triplaneturbo_amd import ops
    from triplaneturbo_amd.registry import BaseModule, __modules__, register
    ref = dict(np.load(os.path.join(golden_dir, "reference_renderer.npz")))
    T = lambda a: torch.from_numpy(np.asarray(a))

    if "fixture-base-renderer" not in __modules__:
        @register("fixture-base-renderer")
        class FixtureBase(BaseModule):
            def configure(self, geometry=None, material=None, background=None):
                pass

            def forward(self, rays_o, rays_d, light_positions, bg_color, **kw):
                s = rays_d.sum(-1, keepdim=True)
                return {"comp_rgb": torch.sin(3.0 * rays_d) + rays_o, "opacity": torch.cos(2.0 * s),
                        "depth": s * s, "not_image": torch.arange(5.0), "scalar": torch.tensor(1.0)}

            def update_step(self, *a, **k):
                pass

    pr = tt.find("patch-renderer")(dict(patch_size=5, global_downsample=3,
                                        base_renderer_type="fixture-base-renderer", base_renderer={}),
                                   geometry=None, material=None, background=None)
    pr.base_renderer.train()
    torch.manual_seed(int(ref["pr_seed"]))
    out = pr(T(ref["pr_rays_o"]).to(dev), T(ref["pr_rays_d"]).to(dev), torch.zeros(2, 3, device=dev), None)
    for key in ("comp_rgb", "opacity", "depth"):
        torch.testing.assert_close(out[key].cpu(), T(ref[f"pr_{key}"]), rtol=1e-6, atol=1e-6)
    assert out["not_image"].shape == (5,) and out["scalar"].ndim == 0  # non-image keys pass through untouched
    # backward: adjoint of upsample + paste, incl. the region the patch overwrites (no gradient to the global render)
    g = torch.Generator().manual_seed(9)
    for (B, h, w, H, W, C, PS, py, px) in ((2, 4, 4, 12, 12, 3, 5, 6, 1), (3, 42, 42, 128, 128, 1, 40, 17, 80),
                                           (1, 7, 5, 23, 16, 2, 4, 0, 12)):
        low = torch.randn(B, h, w, C, generator=g)
        patch = torch.randn(B, PS, PS, C, generator=g)
        go = torch.randn(B, H, W, C, generator=g)
        lo_c, pa_c = low.clone().requires_grad_(True), patch.clone().requires_grad_(True)
        up = torch.nn.functional.interpolate(lo_c.permute(0, 3, 1, 2), (H, W), mode="bilinear").permute(0, 2, 3, 1)
        up = up.clone()
        up[:, py:py + PS, px:px + PS] = pa_c
        gl_ref, gp_ref = torch.autograd.grad((up * go).sum(), [lo_c, pa_c])
        lo_d, pa_d = low.to(dev).requires_grad_(True), patch.to(dev).requires_grad_(True)
        got = ops.patch_composite(lo_d, pa_d, py, px, H, W)
        torch.testing.assert_close(got.cpu(), up.detach(), rtol=1e-6, atol=1e-6)
        gl, gp = torch.autograd.grad((got * go.to(dev)).sum(), [lo_d, pa_d])
        torch.testing.assert_close(gl.cpu(), gl_ref, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(gp.cpu(), gp_ref, rtol=0, atol=0)
        lo_e = low.to(dev).requires_grad_(True)  # global_detach: no gradient to the low-resolution render
        got = ops.patch_composite(lo_e, pa_d, py, px, H, W, detach_low=True)
        assert torch.autograd.grad((got * go.to(dev)).sum(), [lo_e], allow_unused=True)[0] is None


@pytest.mark.parametrize("mode", ["camera", "front", "world"])
def test_fused_composite_matches_torch_reference_ops(dev, mode):
    """tt_composite_fwd / _bwd against the reference's composite written out in torch ops (renderer :433-530, with
    torch.inverse for the world-to-camera rotation), values and gradients incl. d/d background colour, non-orthonormal
    camera matrices, rays with a zero normal sum and disparities on both clamp edges."""
    from triplaneturbo_amd import ops
    g = torch.Generator().manual_seed(31)
    B, rpv, n_view = 4, 37, 2
    n = B * rpv
    op = torch.rand(n, 1, generator=g)
    op[:5] = 0.0
    op[5:9] = 1.0
    dep = torch.rand(n, 1, generator=g) * 4.0
    fg = torch.rand(n, 3, generator=g)
    na = torch.randn(n, 3, generator=g) * op
    bg = torch.rand(n, 3, generator=g)
    _, _, c2w, cd = O.make_cameras(B, 2, 2, azimuth_start_deg=17.0)
    c2w = c2w.clone()
    c2w[:, :3, :3] = c2w[:, :3, :3] @ torch.diag(torch.tensor([1.0, 1.5, 0.8]))
    cd = cd * torch.tensor([1.0, 0.3, 2.5, 1.0])  # far/near straddle the depths: clamp edges are exercised
    outs_g = [torch.randn(n, c, generator=g) for c in (3, 1, 3, 3, 3)]

    def ref(opacity, depth, rgb_fg, nacc, bgc):
        comp_rgb = rgb_fg + bgc * (1.0 - opacity)
        cdv = cd.reshape(-1, 1, 1)
        far, near = cdv + 3 ** 0.5, cdv - 3 ** 0.5
        o3, d3 = opacity.view(B, rpv, 1), depth.view(B, rpv, 1)
        disp = torch.clamp((far - (d3 * o3 + (1 - o3) * far)) / (far - near), 0.0, 1.0).view(n, 1)
        cn = torch.nn.functional.normalize(nacc, dim=-1)
        res = [comp_rgb, disp, cn]
        if mode != "world":
            cc = c2w if mode == "camera" else c2w[0::n_view].repeat_interleave(n_view, 0)
            rot = torch.inverse(cc)[:, :3, :3]
            cam = (cn.view(B, -1, 3) @ rot.permute(0, 2, 1)).view(-1, 3)
            if mode == "camera":
                cam = cam @ torch.diag(torch.tensor([-1.0, 1.0, 1.0]))
                bgn = torch.tensor([0.5, 0.5, 1.0]).expand(n, 3)
                res.append((cam + 1) / 2 * opacity + (1 - opacity) * bgn)
            res.append((cam + 1) / 2 * opacity + (1 - opacity))
        return res

    leaves = [t.clone().requires_grad_(True) for t in (op, dep, fg, na, bg)]
    want = ref(*leaves)
    sel = {"camera": [0, 1, 2, 3, 4], "front": [0, 1, 2, 4], "world": [0, 1, 2]}[mode]
    gw = torch.autograd.grad(sum((w * outs_g[k]).sum() for w, k in zip(want, sel)), leaves)
    dl = [t.clone().to(dev).requires_grad_(True) for t in (op, dep, fg, na, bg)]
    got = ops.composite(dl[0], dl[1], dl[2], dl[3], dl[4], cd.to(dev), c2w.to(dev), rpv, mode, view_group=n_view)
    got = [t for t in got if t is not None]
    for a, b in zip(got, want):
        torch.testing.assert_close(a.cpu(), b.detach(), rtol=2e-5, atol=2e-6)
    gg = torch.autograd.grad(sum((a * outs_g[k].to(dev)).sum() for a, k in zip(got, sel)), dl)
    for name, a, b in zip(("opacity", "depth", "rgb_fg", "normal_acc", "bg"), gg, gw):
        if name == "normal_acc":  # rows with a zero sum: F.normalize's eps branch, gradient x 1e12 on both sides
            ok = na.abs().sum(-1) > 0
            torch.testing.assert_close(a.cpu()[ok], b[ok], rtol=2e-4, atol=2e-5)
        else:
            torch.testing.assert_close(a.cpu(), b, rtol=2e-5, atol=2e-5)
    # constant background colour (3,): its gradient is the sum over rays
    bgc = torch.rand(3, generator=g).to(dev).requires_grad_(True)
    o = ops.composite(dl[0].detach(), dl[1].detach(), dl[2].detach(), dl[3].detach(), bgc, cd.to(dev), c2w.to(dev), rpv,
                      mode, view_group=n_view)[0]
    gb, = torch.autograd.grad((o * outs_g[0].to(dev)).sum(), [bgc])
    torch.testing.assert_close(gb.cpu(), (outs_g[0] * (1 - op)).sum(0), rtol=1e-4, atol=1e-5)


def test_composite_world_mode_without_c2w_and_broadcast_camera_distance(dev):
    """normal_direction='world' needs no c2w (the reference only touches it for 'camera' / 'front', renderer :478-530)
    and a 1-element camera_distances broadcasts over the views as it does in the reference's arithmetic (:452-456);
    anything else that does not match the number of views is refused instead of being read out of bounds."""
    from triplaneturbo_amd import ops
    g = torch.Generator().manual_seed(5)
    B, rpv = 3, 11
    n = B * rpv
    op, dep = torch.rand(n, 1, generator=g), torch.rand(n, 1, generator=g) * 3
    fg, na, bg = torch.rand(n, 3, generator=g), torch.randn(n, 3, generator=g), torch.rand(3, generator=g)
    cd1 = torch.tensor([1.7])
    got = ops.composite(op.to(dev), dep.to(dev), fg.to(dev), na.to(dev), bg.to(dev), cd1.to(dev), None, rpv, "world")
    want = ops.composite(op.to(dev), dep.to(dev), fg.to(dev), na.to(dev), bg.to(dev), cd1.expand(B).contiguous().to(dev),
                         torch.eye(4).expand(B, 4, 4).contiguous().to(dev), rpv, "world")
    for a, b in zip(got[:3], want[:3]):
        assert torch.equal(a, b)
    far, near = 1.7 + 3 ** 0.5, 1.7 - 3 ** 0.5
    disp = torch.clamp((far - (dep * op + (1 - op) * far)) / (far - near), 0.0, 1.0)
    torch.testing.assert_close(got[1].cpu(), disp, rtol=2e-5, atol=2e-6)
    assert got[3] is None and got[4] is None
    with pytest.raises(ValueError):
        ops.composite(op.to(dev), dep.to(dev), fg.to(dev), na.to(dev), bg.to(dev), cd1.expand(2).contiguous().to(dev),
                      None, rpv, "world")
    with pytest.raises(ValueError):  # camera-space normals need the cameras
        ops.composite(op.to(dev), dep.to(dev), fg.to(dev), na.to(dev), bg.to(dev), cd1.to(dev), None, rpv, "camera")
    with pytest.raises(ValueError):
        ops.composite(op.to(dev), dep.to(dev), fg.to(dev), na.to(dev), bg.to(dev), cd1.to(dev),
                      torch.eye(4).expand(2, 4, 4).contiguous().to(dev), rpv, "camera")
