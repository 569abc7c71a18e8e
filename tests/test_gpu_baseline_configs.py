"""GPU tests at the sizes of BASELINE.json configs[2], [3] (one rank's share) and [4] (its HIP share): full-size runs
with size-independent properties, plus oracle parity -- outputs AND gradients w.r.t. the 256^2 planes and the six
matrices -- on a scattered subset of rays / points (per-ray results do not depend on the other rays and gradients are
linear in rays, which the full-size linearity checks tie together).

  configs[2]  8 prompts x 4 views, 128x128 rays through PatchRenderer (42^2 global + 40^2 patch rays per view), importance
              sampling 128 + 64 -> 193 samples (configs/TriplaneTurbo_v1.yaml:8-9,133-150), planes 256^2
  configs[3]  64 prompts sharded 8-way: one rank renders 8 prompts x 256x256 rays x 128 samples
  configs[4]  text -> mesh: forward_field on the exporter's 160^3 grid + vertex colouring of ~300 k points
              (triplaneturbo_executable/utils/mesh_exporter.py:78-183); SD-UNet and marching cubes are stock / out of scope
"""
import pytest
import torch

import triplaneturbo_amd as tt
from oracle import cpu_ref as O

from parity import check_grads, check_outputs, rel, report

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda", 0)


def _modules(dev, deform=False):
    torch.manual_seed(0)
    g = tt.find("few-step-triplane-dual-stable-diffusion")({"isosurface_deformable_grid": deform}).to(dev)
    m = tt.find("no-material")({})
    b = tt.find("solid-color-background")({})
    base = dict(estimator="importance", trainable_variance=False, learned_variance_init=0.4605, num_samples_per_ray=64,
                num_samples_per_ray_importance=128, near_plane=0.1, far_plane=4.0,
                rgb_grad_shrink=[0, 1, 0.01, 20000], randomized=True)
    return g, m, b, base


PROJ_KEYS = (("comp_rgb", 3), ("opacity", 1), ("depth", 1), ("disparity", 1), ("comp_normal_cam_vis", 3))


def _subset_parity(case, dev, ops, functional, cache2, sw, fw, ro, rd, ts, te, c2w, cd, rc_kwargs, seed):
    """HIP vs fp32 / fp64 oracle on a small ray set: cache2 (P,6,32,R,R) cpu; ro/rd (B,1,n,3) cpu with B = P * views;
    ts/te (B*n, S) cpu.  Outputs and gradients of the G6-form loss."""
    g = torch.Generator().manual_seed(seed)
    B, Hh, Ww, _ = ro.shape
    proj = {k: torch.randn(B, Hh, Ww, c, generator=g) for k, c in PROJ_KEYS}
    bg = torch.ones(3)

    def hip():
        c = cache2.to(dev).requires_grad_(True)
        s = [w.to(dev).requires_grad_(True) for w in sw]
        f = [w.to(dev).requires_grad_(True) for w in fw]
        rc = ops.RenderConfig(**rc_kwargs)
        out = functional.volume_render(c, s, f, ro.to(dev), rd.to(dev), ts.to(dev), te.to(dev), bg.to(dev),
                                       cd.to(dev), c2w.to(dev), rc, training=True)
        loss = O.synthetic_loss(out, {k: v.to(dev) for k, v in proj.items()})
        return out, torch.autograd.grad(loss, [c] + s + f)

    def oracle(d):
        c = cache2.to(d).requires_grad_(True)
        s = [w.to(d).requires_grad_(True) for w in sw]
        f = [w.to(d).requires_grad_(True) for w in fw]
        kw = {k: v for k, v in rc_kwargs.items() if k in ("inv_std", "rgb_grad_shrink", "cos_anneal_ratio")}
        out = O.render(c, s, f, ro.to(d), rd.to(d), ts.to(d), te.to(d), bg.to(d), cd.to(d), c2w.to(d), **kw)
        loss = O.synthetic_loss(out, {k: v.to(d) for k, v in proj.items()})
        return out, torch.autograd.grad(loss, [c] + s + f)

    out_h, g_h = hip()
    o32, g32 = oracle(torch.float32)
    o64, g64 = oracle(torch.float64)
    check_outputs(case + " outputs", out_h, o32, o64,
                  ("comp_rgb", "opacity", "depth", "z_variance", "disparity", "comp_normal_cam_vis", "weights", "sdf",
                   "features", "sdf_grad"))
    return check_grads(case + " gradients", g_h, g32, g64)


# --------------------------------------------------------------------------------------------------------------
def test_config2_eight_prompts_four_views_patch_renderer(dev):
    from triplaneturbo_amd import functional, ops
    g, m, b, base = _modules(dev)
    r = tt.find("patch-renderer")({"patch_size": 40, "global_downsample": 3,
                                   "base_renderer_type": "generative-space-sdf-volume-renderer",
                                   "base_renderer": base}, geometry=g, material=m, background=b).to(dev)
    r.train()
    r.update_step(0, 1000)
    gen = torch.Generator().manual_seed(2)
    P, n_view, R = 8, 4, 256
    cache = (torch.randn(P, 6, 32, R, R, generator=gen) * 0.5).to(dev).requires_grad_(True)
    ro, rd, c2w, cd = O.make_cameras(P * n_view, 128, 128)
    kw = dict(space_cache=cache, text_embed=torch.zeros(P, 77, 1024), camera_distances=cd.to(dev), c2w=c2w.to(dev))
    torch.manual_seed(3)
    out = r(ro.to(dev), rd.to(dev), None, torch.ones(3, device=dev), **kw)
    loss = out["comp_rgb"].mean() + (out["opacity"] ** 2 + 0.01).sqrt().mean() + \
        ((out["sdf_grad"].norm(dim=-1) - 1) ** 2).mean() + out["comp_normal_cam_vis"].mean()
    loss.backward()
    S = 193
    n_glob = P * n_view * 42 * 42
    assert out["comp_rgb"].shape == (32, 128, 128, 3) and out["weights"].shape == (n_glob * S, 1)
    w = out["weights"].view(n_glob, S)
    assert torch.isfinite(w).all() and (w >= 0).all() and (w.sum(1) <= 1 + 1e-4).all()
    assert (out["t_intervals"] >= 0).all()  # importance-sampled intervals are sorted
    assert torch.isfinite(loss) and torch.isfinite(cache.grad).all()
    per_prompt = cache.grad.flatten(1).abs().sum(1)
    assert (per_prompt > 0).all()  # every prompt's planes received gradient from its own 4 views
    for p_ in g.parameters():
        assert torch.isfinite(p_.grad).all() and p_.grad.abs().sum() > 0

    # ---- oracle parity on rays of this very workload: 3 global-render rays of every view of prompts 0 and 7,
    # with the intervals the HIP importance sampler placed for them ----
    base_r = r.base_renderer
    F = torch.nn.functional
    lo_o = F.interpolate(ro.permute(0, 3, 1, 2), (42, 42), mode="bilinear").permute(0, 2, 3, 1).contiguous()
    lo_d = F.interpolate(rd.permute(0, 3, 1, 2), (42, 42), mode="bilinear").permute(0, 2, 3, 1).contiguous()
    ts, te = base_r.sample(cache.detach(), lo_o.to(dev), lo_d.to(dev), generator=torch.Generator(dev).manual_seed(5))
    assert ts.shape == (n_glob, S)
    views = [0, 1, 2, 3, 28, 29, 30, 31]
    pix = torch.tensor([42 * 21 + 20, 42 * 17 + 26, 42 * 30 + 9])  # through / near the object, and a grazing one
    sel = torch.cat([v * 42 * 42 + pix for v in views])
    sw = [w_.detach().cpu() for w_ in g.mlp_weights()[0]]
    fw = [w_.detach().cpu() for w_ in g.mlp_weights()[1]]
    rows = _subset_parity("configs[2] subset (prompts 0,7; 24 rays x 193 importance samples; planes 256^2)", dev, ops,
                          functional, cache.detach().cpu()[[0, 7]], sw, fw,
                          lo_o.reshape(-1, 3)[sel].view(8, 1, 3, 3), lo_d.reshape(-1, 3)[sel].view(8, 1, 3, 3),
                          ts.cpu()[sel], te.cpu()[sel], c2w[views], cd[views],
                          dict(inv_std=base_r._inv_std_value(), rgb_grad_shrink=float(base_r.rgb_grad_shrink),
                               tile_sb=8), seed=6)
    print(rows)


def test_config3_one_rank_share_eight_prompts_256x256(dev):
    from triplaneturbo_amd import functional, ops
    P, R, Hh, Ww, S = 8, 256, 256, 256, 128
    gen = torch.Generator().manual_seed(4)
    cache = (torch.randn(P, 6, 32, R, R, generator=gen) * 0.5).to(dev)
    sw = O.init_mlp_weights([32, 64, 64, 1], gen)
    fw = O.init_mlp_weights([96, 64, 64, 3], gen)
    ro, rd, c2w, cd = O.make_cameras(P, Hh, Ww)  # one view per prompt, azimuths spread over the circle
    n = P * Hh * Ww
    ts, te = O.uniform_intervals(Hh * Ww, S, 0.1, 4.0)
    ts, te = ts.to(dev).repeat(P, 1), te.to(dev).repeat(P, 1)
    rof, rdf = ro.reshape(-1, 3).to(dev), rd.reshape(-1, 3).to(dev)
    rc = ops.RenderConfig()
    pr = torch.randn(n, 3, generator=gen).to(dev)
    swd, fwd = [w.to(dev) for w in sw], [w.to(dev) for w in fw]

    def grads(prompts):
        """render + backward of the rays of a contiguous range of prompts (their planes only)"""
        lo, hi = prompts[0] * Hh * Ww, (prompts[-1] + 1) * Hh * Ww
        c = cache[prompts[0]:prompts[-1] + 1].clone().requires_grad_(True)
        s = [w.clone().requires_grad_(True) for w in swd]
        f = [w.clone().requires_grad_(True) for w in fwd]
        r = ops.render_samples(c, s, f, rof[lo:hi], rdf[lo:hi], ts[lo:hi], te[lo:hi], Hh * Ww, rc, image_w=Ww)
        loss = (r["rgb_fg"] * pr[lo:hi]).sum() + r["opacity"].sum() + ((r["sdf_grad"].norm(dim=-1) - 1) ** 2).sum()
        return r, torch.autograd.grad(loss, [c] + s + f)

    r, full = grads(list(range(P)))
    assert all(torch.isfinite(v).all() for v in r.values())
    w = r["weights"].view(n, S)
    torch.testing.assert_close(w.sum(1, keepdim=True), r["opacity"], rtol=1e-5, atol=1e-6)
    assert (r["opacity"] >= 0).all() and (r["opacity"] <= 1 + 1e-5).all()
    # prompts are independent units: the batch of 8 equals two batches of 4 (planes gradients concatenate, MLP
    # gradients add) -- what the 8-way shard of configs[3] relies on
    _, a = grads([0, 1, 2, 3])
    _, b = grads([4, 5, 6, 7])
    assert rel(torch.cat([a[0], b[0]]), full[0]) < 2e-5
    for x, y, z in zip(full[1:], a[1:], b[1:]):
        assert rel(y + z, x) < 2e-5
    report("configs[3] share: 8 prompts x 65536 rays x 128 samples, batch == sum of two half batches",
           {"planes": rel(torch.cat([a[0], b[0]]), full[0]), "sdf.w2": rel(a[2] + b[2], full[2])})
    # ---- oracle parity (outputs + gradients at 256^2 planes) on 32 scattered rays of prompts 0 and 7 ----
    pix = torch.arange(0, Hh * Ww, 4099)[:16] + 17
    sel = torch.cat([pix, 7 * Hh * Ww + pix])
    rows = _subset_parity("configs[3] subset (prompts 0,7; 32 rays x 128 samples; planes 256^2)", dev, ops, functional,
                          cache.cpu()[[0, 7]], sw, fw, rof.cpu()[sel].view(2, 1, 16, 3), rdf.cpu()[sel].view(2, 1, 16, 3),
                          ts.cpu()[sel], te.cpu()[sel], c2w[[0, 7]], cd[[0, 7]],
                          dict(inv_std=100.0, rgb_grad_shrink=1.0), seed=7)
    print(rows)


def test_config4_field_query_160_cubed_and_vertex_colouring(dev):
    g, *_ = _modules(dev, deform=True)
    gen = torch.Generator().manual_seed(8)
    cache = (torch.randn(1, 6, 32, 256, 256, generator=gen) * 0.5)
    res = 160
    lin = torch.linspace(-1.0, 1.0, res)  # isosurface_bbox hard-coded to [-1, 1] (mesh_exporter.py:98-102)
    grid = torch.stack(torch.meshgrid(lin, lin, lin, indexing="ij"), -1).reshape(1, -1, 3)
    with torch.no_grad():
        sdf, deform = g.forward_field(grid.to(dev), cache.to(dev))
    assert sdf.shape == (1, res ** 3, 1) and deform.shape == (1, res ** 3, 3)
    assert torch.isfinite(sdf).all() and torch.isfinite(deform).all()
    assert (sdf < 0).any() and (sdf > 0).any()  # the sphere bias puts a level set inside the box
    sw = [w.detach().cpu() for w in g.mlp_weights()[0]]
    fw = [w.detach().cpu() for w in g.mlp_weights()[1]]
    dw = [w.detach().cpu() for w in g.deformation_network.weights()]
    sel = torch.arange(0, res ** 3, 1999)[:2048]
    d = torch.float64
    want64 = O.geometry_forward(grid[:, sel].to(d), cache.to(d), [w.to(d) for w in sw], [w.to(d) for w in fw],
                                output_normal=False)
    want32 = O.geometry_forward(grid[:, sel], cache, sw, fw, output_normal=False)
    def64 = O.vanilla_mlp(want64["enc_geo"], [w.to(d) for w in dw])
    def32 = O.vanilla_mlp(want32["enc_geo"], dw)
    got = {"sdf": sdf[0, sel.to(dev)], "deformation": deform[0, sel.to(dev)]}
    check_outputs("configs[4] forward_field 160^3 (2048-point subset, planes 256^2)", got,
                  {"sdf": want32["sdf"], "deformation": def32}, {"sdf": want64["sdf"], "deformation": def64},
                  ("sdf", "deformation"))
    # vertex colouring: ~300 k surface points (export, few_step...:402-430)
    n_v = 300_000
    v = torch.nn.functional.normalize(torch.randn(n_v, 3, generator=gen), dim=-1) * (0.5 + 0.05 * torch.randn(
        n_v, 1, generator=gen))
    with torch.no_grad():
        col = g.export(v.to(dev), cache.to(dev))["features"]
    assert col.shape == (n_v, 3) and torch.isfinite(col).all()
    selv = torch.arange(0, n_v, 293)[:1024]
    c64 = O.geometry_forward(v[selv][None].to(d), cache.to(d), [w.to(d) for w in sw], [w.to(d) for w in fw],
                             output_normal=False)["features"]
    c32 = O.geometry_forward(v[selv][None], cache, sw, fw, output_normal=False)["features"]
    check_outputs("configs[4] export vertex colours (1024 of 300 k points)", {"features": col[selv.to(dev)]},
                  {"features": c32}, {"features": c64}, ("features",))
