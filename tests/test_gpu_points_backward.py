"""Backward of the per-point queries (tt_points_bwd_geo / tt_points_bwd_tex) against autograd of the oracle:
geometry.forward under autograd (sdf, sdf_grad second order, features) and forward_field (sdf + deformation head) --
SURVEY 8(f) ranks 1 and 3 (the training-time callers are the mesh-rasterize renderer's grid query
generative_space_mesh_rasterize_renderer.py:428-452 and per-pixel decode :321-376).
Tolerance as everywhere: HIP must be as close to the fp64 oracle as the fp32 oracle is (x3) or 1e-4 in norm."""
import pytest
import torch

import triplaneturbo_amd as tt
from oracle import cpu_ref as O

from parity import PRECISIONS  # noqa: E402

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def _check(g_hip, g32, g64, names):
    for n, a, b, c in zip(names, g_hip, g32, g64):
        e_hip, e_cpu = _rel(a.cpu(), c), _rel(b, c)
        assert e_hip <= max(1e-4, 3 * e_cpu), (n, e_hip, e_cpu)


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("output_normal", [True, False])
def test_geometry_forward_is_differentiable(output_normal, precision):
    dev = torch.device("cuda", 0)
    torch.manual_seed(11)
    g = tt.find("few-step-triplane-dual-stable-diffusion")({}).to(dev)
    g.precision = precision
    gen = torch.Generator().manual_seed(12)
    P, vpp, N, R = 2, 2, 333, 32
    cache = torch.randn(P, 6, 32, R, R, generator=gen) * 0.5
    pts = torch.rand(P * vpp, N, 3, generator=gen) * 2.2 - 1.1  # some outside the box (zeros padding)
    sw = [w.detach().cpu() for w in g.sdf_network.weights()]
    fw = [w.detach().cpu() for w in g.feature_network.weights()]
    keys = ("sdf", "features") + (("sdf_grad", "normal") if output_normal else ())
    proj = {k: torch.randn(P * vpp * N, 1 if k == "sdf" else 3, generator=gen) for k in keys}

    def oracle(dt):
        c = cache.to(dt).requires_grad_(True)
        ws = [w.to(dt).requires_grad_(True) for w in sw + fw]
        o = O.geometry_forward(pts.to(dt), c.repeat_interleave(vpp, 0), ws[:3], ws[3:], output_normal=output_normal,
                               create_graph=True)
        loss = sum((o[k] * proj[k].to(dt)).sum() for k in keys)
        return o, torch.autograd.grad(loss, [c] + ws)

    o32, g32 = oracle(torch.float32)
    o64, g64 = oracle(torch.float64)
    c = cache.to(dev).requires_grad_(True)
    out = g(pts.to(dev), c, output_normal=output_normal)
    for k in keys:
        e_hip = (out[k].detach().cpu().double() - o64[k].detach()).abs().max().item()
        e_cpu = (o32[k].detach().double() - o64[k].detach()).abs().max().item()
        assert e_hip <= max(4 * e_cpu, 2e-5), (k, e_hip, e_cpu)
    loss = sum((out[k] * proj[k].to(dev)).sum() for k in keys)
    params = [c] + list(g.sdf_network.weights()) + list(g.feature_network.weights())
    g_hip = torch.autograd.grad(loss, params)
    _check(g_hip, g32, g64, ["planes", "w1", "w2", "w3", "v1", "v2", "v3"])
    # inference-style call leaves no graph
    with torch.no_grad():
        assert not g(pts.to(dev), c, output_normal=output_normal)["sdf"].requires_grad


@pytest.mark.parametrize("precision", PRECISIONS)
def test_forward_field_is_differentiable(precision):
    dev = torch.device("cuda", 0)
    torch.manual_seed(13)
    g = tt.find("few-step-triplane-dual-stable-diffusion")({"isosurface_deformable_grid": True}).to(dev)
    g.precision = precision
    gen = torch.Generator().manual_seed(14)
    R = 32
    cache = torch.randn(1, 6, 32, R, R, generator=gen) * 0.5
    lin = torch.linspace(-1.05, 1.05, 11)
    pts = torch.stack(torch.meshgrid(lin, lin, lin, indexing="ij"), -1).reshape(1, -1, 3)
    n = pts.shape[1]
    sw = [w.detach().cpu() for w in g.sdf_network.weights()]
    fw = [w.detach().cpu() for w in g.feature_network.weights()]
    dw = [w.detach().cpu() for w in g.deformation_network.weights()]
    p_sdf, p_def = torch.randn(n, 1, generator=gen), torch.randn(n, 3, generator=gen)

    def oracle(dt):
        c = cache.to(dt).requires_grad_(True)
        ws = [w.to(dt).requires_grad_(True) for w in sw + dw]
        o = O.geometry_forward(pts.to(dt), c, ws[:3], [w.to(dt) for w in fw], output_normal=False)
        deform = O.vanilla_mlp(o["enc_geo"], ws[3:])
        loss = (o["sdf"] * p_sdf.to(dt)).sum() + (deform * p_def.to(dt)).sum()
        return torch.autograd.grad(loss, [c] + ws)

    g32, g64 = oracle(torch.float32), oracle(torch.float64)
    c = cache.to(dev).requires_grad_(True)
    sdf, deform = g.forward_field(pts.to(dev), c)
    assert sdf.requires_grad and deform.requires_grad
    loss = (sdf.reshape(-1, 1) * p_sdf.to(dev)).sum() + (deform.reshape(-1, 3) * p_def.to(dev)).sum()
    params = [c] + list(g.sdf_network.weights()) + list(g.deformation_network.weights())
    g_hip = torch.autograd.grad(loss, params)
    _check(g_hip, g32, g64, ["planes", "w1", "w2", "w3", "d1", "d2", "d3"])
    assert g_hip[0][:, 3:].abs().max().item() == 0.0  # texture planes are not touched by the field query


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("output_normal", [True, False])
def test_gradient_wrt_the_query_points(output_normal, precision):
    """the raster renderer decodes positions interpolated from differentiable mesh vertices, so
    d loss / d points must flow (first order through sdf / features, second order through sdf_grad / normal: K1's
    grad_grid output, gridsample_cuda.cu:196-208).  Against autograd of the oracle with points.requires_grad_()."""
    from parity import report
    dev = torch.device("cuda", 0)
    torch.manual_seed(21)
    g = tt.find("few-step-triplane-dual-stable-diffusion")({}).to(dev)
    g.precision = precision
    gen = torch.Generator().manual_seed(22)
    P, vpp, N, R = 2, 2, 257, 48
    cache = torch.randn(P, 6, 32, R, R, generator=gen) * 0.5
    pts = torch.rand(P * vpp, N, 3, generator=gen) * 2.2 - 1.1  # some outside the box (zeros padding)
    sw = [w.detach().cpu() for w in g.sdf_network.weights()]
    fw = [w.detach().cpu() for w in g.feature_network.weights()]
    keys = ("sdf", "features") + (("sdf_grad", "normal") if output_normal else ())
    proj = {k: torch.randn(P * vpp * N, 1 if k == "sdf" else 3, generator=gen) for k in keys}

    def oracle(dt):
        x = pts.to(dt).requires_grad_(True)
        c = cache.to(dt).requires_grad_(True)
        o = O.geometry_forward(x, c.repeat_interleave(vpp, 0), [w.to(dt) for w in sw], [w.to(dt) for w in fw],
                               output_normal=output_normal, create_graph=True)
        loss = sum((o[k] * proj[k].to(dt)).sum() for k in keys)
        return torch.autograd.grad(loss, [x, c])

    gx32, gc32 = oracle(torch.float32)
    gx64, gc64 = oracle(torch.float64)
    x = pts.to(dev).requires_grad_(True)
    c = cache.to(dev).requires_grad_(True)
    out = g(x, c, output_normal=output_normal)
    loss = sum((out[k] * proj[k].to(dev)).sum() for k in keys)
    gx, gc = torch.autograd.grad(loss, [x, c])
    e_hip, e_cpu, e_32 = _rel(gx.cpu(), gx64), _rel(gx32, gx64), _rel(gx.cpu(), gx32)
    report(f"d/d points of geometry.forward (output_normal={output_normal}, precision={precision})",
           {"hip_vs_fp64": e_hip, "fp32_vs_fp64": e_cpu, "hip_vs_fp32": e_32})
    assert e_hip <= max(1e-4, 3 * e_cpu), (e_hip, e_cpu)
    assert e_32 <= 1e-4, e_32
    _check([gc], [gc32], [gc64], ["planes"])
    # points only (frozen planes and weights): the plane / weight backward kernels are skipped altogether
    for w_ in g.parameters():
        w_.requires_grad_(False)
    x2 = pts.to(dev).requires_grad_(True)
    out2 = g(x2, cache.to(dev), output_normal=output_normal)
    gx2, = torch.autograd.grad(sum((out2[k] * proj[k].to(dev)).sum() for k in keys), [x2])
    torch.testing.assert_close(gx2, gx, rtol=1e-6, atol=1e-7)
