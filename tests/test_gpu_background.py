"""The hashgrid + hypernet background (reference
multi_prompt_neural_environment_hashgrid_map_background.py, tiny-cuda-nn HashGrid) on the HIP kernels
tt_hashgrid_fwd / tt_hashgrid_bwd, against the oracle restatement (tcnn is CUDA-only and un-vendored: parity with
tcnn itself is unpinned; tcnn computes this op in fp16, we compute it in fp32)."""
import pytest
import torch

import triplaneturbo_amd as tt
from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu
NAME = "multi-prompt-neural-hashgrid-environment-map-background"


def _rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("cfg", [
    dict(n_levels=8, n_features=2, log2_hashmap_size=19, base_resolution=4, per_level_scale=1.8114473285278132),
    dict(n_levels=5, n_features=4, log2_hashmap_size=10, base_resolution=3, per_level_scale=2.0),
    dict(n_levels=3, n_features=1, log2_hashmap_size=14, base_resolution=16, per_level_scale=1.5),
])
def test_hashgrid_encode_and_table_gradient(cfg):
    from triplaneturbo_amd.background import HashGrid
    g = torch.Generator().manual_seed(1)
    enc = HashGrid(3, {"otype": "HashGrid", "n_levels": cfg["n_levels"], "n_features_per_level": cfg["n_features"],
                       "log2_hashmap_size": cfg["log2_hashmap_size"], "base_resolution": cfg["base_resolution"],
                       "per_level_scale": cfg["per_level_scale"]}).cuda()
    _, total = O.hashgrid_levels(cfg["n_levels"], cfg["log2_hashmap_size"], cfg["base_resolution"],
                                 cfg["per_level_scale"])
    assert enc.params.numel() == total * cfg["n_features"]
    assert enc.params.abs().max().item() <= 1e-4  # tcnn's U(-1e-4, 1e-4) initialisation
    with torch.no_grad():
        enc.params.copy_(torch.randn(enc.params.numel(), generator=g))
    x = torch.rand(777, 3, generator=g)
    x[:5] = torch.tensor([[0.0, 0.0, 0.0], [1.0, 1.0, 1.0], [0.5, 0.5, 0.5], [1.0, 0.0, 0.25], [0.0, 1.0, 0.75]])
    proj = torch.randn(777, cfg["n_levels"] * cfg["n_features"], generator=g)
    p64 = enc.params.detach().cpu().double().requires_grad_(True)
    want = O.hashgrid_encode(x.double(), p64, **cfg)
    (g64,) = torch.autograd.grad((want * proj.double()).sum(), p64)
    p32 = enc.params.detach().cpu().requires_grad_(True)
    w32 = O.hashgrid_encode(x, p32, **cfg)
    (g32,) = torch.autograd.grad((w32 * proj).sum(), p32)
    out = enc(x.cuda())
    e_hip = (out.detach().cpu().double() - want.detach()).abs().max().item()
    e_cpu = (w32.detach().double() - want.detach()).abs().max().item()
    assert e_hip <= max(4 * e_cpu, 1e-6), (e_hip, e_cpu)
    (g_hip,) = torch.autograd.grad((out * proj.cuda()).sum(), enc.params)
    assert _rel(g_hip.cpu(), g64) <= max(1e-5, 3 * _rel(g32, g64))


def test_background_module_matches_oracle_and_reference_layout():
    torch.manual_seed(2)
    bg = tt.find(NAME)({"color_activation": "sigmoid-mipnerf", "eval_color": [1.0, 1.0, 1.0]}).cuda()
    keys = list(bg.state_dict().keys())
    assert keys == ["encoding.encoding.encoding.params", "hypernet.layers.0.weight", "hypernet.layers.1.weight",
                    "hypernet.layers.1.bias", "hypernet.layers.3.weight", "hypernet.layers.3.bias"]
    assert bg.hypernet.layers[3].weight.shape == (16 * 64 + 64 * 3, 64) and bg.enabling_hypernet
    gen = torch.Generator().manual_seed(3)
    with torch.no_grad():
        bg.encoding.encoding.encoding.params.copy_(torch.randn(bg.encoding.encoding.encoding.params.numel(),
                                                               generator=gen) * 0.5)
    P, nv, Hh, Ww = 2, 2, 5, 7
    _, rd, _, _ = O.make_cameras(P * nv, Hh, Ww)
    text = torch.randn(P, 1024, generator=gen)
    bg.eval()
    assert torch.equal(bg(rd.cuda(), text.cuda()).cpu(), torch.ones(P * nv, Hh, Ww, 3))  # eval_color
    bg.train()
    col = bg(rd.cuda(), text.cuda())
    sd = {k: v.detach().cpu() for k, v in bg.state_dict().items()}
    hyper = [sd["hypernet.layers.0.weight"], sd["hypernet.layers.1.weight"], sd["hypernet.layers.1.bias"],
             sd["hypernet.layers.3.weight"], sd["hypernet.layers.3.bias"]]
    want = O.hypernet_background(rd.double(), text.double(), sd["encoding.encoding.encoding.params"].double(),
                                 [h.double() for h in hyper])
    assert col.shape == (P * nv, Hh, Ww, 3)
    torch.testing.assert_close(col.detach().cpu().double(), want, rtol=1e-4, atol=1e-5)
    # gradients reach the table and the hyper-network
    col.square().mean().backward()
    for n_, p_ in bg.named_parameters():
        assert p_.grad is not None and torch.isfinite(p_.grad).all() and p_.grad.abs().sum() > 0, n_


def test_renderer_composites_the_hypernet_background():
    dev = torch.device("cuda", 0)
    torch.manual_seed(4)
    g = tt.find("few-step-triplane-dual-stable-diffusion")({}).to(dev)
    bg = tt.find(NAME)({"color_activation": "sigmoid-mipnerf"}).to(dev)
    r = tt.find("generative-space-sdf-volume-renderer")(
        dict(estimator="importance", trainable_variance=False, learned_variance_init=0.4605, num_samples_per_ray=16,
             num_samples_per_ray_importance=32, near_plane=0.1, far_plane=4.0),
        geometry=g, material=tt.find("no-material")({}), background=bg).to(dev)
    r.train()
    gen = torch.Generator().manual_seed(5)
    cache = (torch.randn(1, 6, 32, 32, 32, generator=gen) * 0.5).to(dev).requires_grad_(True)
    ro, rd, c2w, cd = O.make_cameras(2, 6, 8)
    text = torch.randn(1, 1024, generator=gen).to(dev)
    out = r(ro.to(dev), rd.to(dev), None, None, space_cache=cache, text_embed=text, camera_distances=cd.to(dev),
            c2w=c2w.to(dev))
    want_bg = bg(rd.to(dev), text)
    torch.testing.assert_close(out["comp_rgb_bg"], want_bg)
    torch.testing.assert_close(out["comp_rgb"], out["comp_rgb_fg"] + want_bg * (1 - out["opacity"]))
    out["comp_rgb"].mean().backward()
    assert bg.encoding.encoding.encoding.params.grad.abs().sum() > 0 and cache.grad.abs().sum() > 0
