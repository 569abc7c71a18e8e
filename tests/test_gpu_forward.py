"""GPU parity tests (forward): HIP kernels through the C ABI vs the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import cpu_ref as O

from parity import PRECISIONS  # noqa: E402

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.asarray(a))


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from triplaneturbo_amd import ops as _ops
    return _ops


def _scene(seed, P, R, n_view, Hh, Ww, S, near=0.1, far=4.0, scale=0.5):
    g = torch.Generator().manual_seed(seed)
    cache = torch.randn(P, 6, 32, R, R, generator=g) * scale
    sw = O.init_mlp_weights([32, 64, 64, 1], g)
    fw = O.init_mlp_weights([96, 64, 64, 3], g)
    ro, rd, c2w, cd = O.make_cameras(P * n_view, Hh, Ww)
    n_rays = P * n_view * Hh * Ww
    ts, te = O.uniform_intervals(n_rays, S, near, far)
    return cache, sw, fw, ro, rd, c2w, cd, ts, te


@pytest.mark.parametrize("R", [16, 18, 48, 52, 256])  # 18: the scalar kernel (W % 4 != 0); 52: row stride padding
def test_planes_pack_is_rotate_v1_channels_last(ops, R):
    g = torch.Generator().manual_seed(0)
    cache = torch.randn(2, 6, 32, R, R, generator=g)
    packed = ops.planes_pack(cache.cuda()).cpu()
    want = O.rotate_planes_v1(cache).permute(0, 1, 3, 4, 2).contiguous()
    assert torch.equal(packed, want)
    # unpack_grad is the exact transpose (a permutation): round trip is the identity
    back = ops.planes_unpack_grad(packed.cuda()).cpu()
    assert torch.equal(back, cache)


def test_planes_unpack_grad_sums_the_privatised_copies(ops):
    g = torch.Generator().manual_seed(1)
    copies = torch.randn(3, 1, 6, 24, 24, 32, generator=g)
    got = ops.planes_unpack_grad(copies.cuda()).cpu()  # 6-d input: (copies, P, 6, H, W, 32)
    one = [ops.planes_unpack_grad(copies[k].contiguous().cuda()).cpu() for k in range(3)]
    torch.testing.assert_close(got, (one[0] + one[1]) + one[2], rtol=0, atol=0)


def test_query_points_matches_reference_golden(ops, golden_dir):
    """tests/golden/reference_ops.npz was produced by the IMPORTED reference functions."""
    ref = dict(np.load(os.path.join(golden_dir, "reference_ops.npz")))
    sw = [T(ref[f"g2_sdf_w{i}"]).cuda() for i in range(3)]
    fw = [T(ref[f"g2_feat_w{i}"]).cuda() for i in range(3)]
    packed = ops.planes_pack(T(ref["g3_cache"]).cuda())
    pts = T(ref["g3_pts"]).cuda()
    sdf, grad, feat = ops.query_points(packed, sw, fw, pts)
    B, N = ref["g3_pts"].shape[:2]
    torch.testing.assert_close(sdf.cpu().view(B, N, 1), T(ref["g4_sdf"]), rtol=1e-5, atol=2e-6)
    torch.testing.assert_close(feat.cpu().view(B, N, 3), T(ref["g4_features"]), rtol=1e-5, atol=2e-6)
    torch.testing.assert_close(grad.cpu().view(B, N, 3), T(ref["g4_sdf_grad"]), rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("R,N", [(16, 33), (48, 777), (128, 5000), (256, 4096)])  # 48: not a power of two
def test_query_points_matches_oracle(ops, R, N, precision):
    g = torch.Generator().manual_seed(R + N)
    P, n_view = 2, 2
    cache = torch.randn(P, 6, 32, R, R, generator=g) * 0.5
    sw = O.init_mlp_weights([32, 64, 64, 1], g)
    fw = O.init_mlp_weights([96, 64, 64, 3], g)
    pts = torch.rand(P * n_view, N, 3, generator=g) * 2.4 - 1.2  # includes out-of-box points
    want = O.geometry_forward(pts, cache.repeat_interleave(n_view, 0), sw, fw, output_normal=True)
    want64 = O.geometry_forward(pts.double(), cache.double().repeat_interleave(n_view, 0), [w.double() for w in sw],
                                [w.double() for w in fw], output_normal=True)
    packed = ops.planes_pack(cache.cuda())
    sdf, grad, feat = ops.query_points(packed, [w.cuda() for w in sw], [w.cuda() for w in fw], pts.cuda(),
                                       views_per_prompt=n_view, precision=precision)
    for name, got in (("sdf", sdf), ("sdf_grad", grad), ("features", feat)):
        got = got.cpu()
        w32, w64 = want[name], want64[name]
        err_hip = (got.double() - w64).abs().max().item()
        err_cpu = (w32.double() - w64).abs().max().item()
        scale = w64.abs().max().item()
        # the HIP result must be as close to exact arithmetic as the fp32 CPU restatement is (x4 slack), and
        # within 1e-5 relative of full scale.
        assert err_hip <= max(4 * err_cpu, 1e-5 * scale), (name, err_hip, err_cpu, scale)
    # sdf-only path (forward_sdf, few_step...:353-373)
    sdf2, g2, f2 = ops.query_points(packed, [w.cuda() for w in sw], None, pts.cuda(), views_per_prompt=n_view,
                                    need_normal=False, need_features=False, precision=precision)
    assert g2 is None and f2 is None
    torch.testing.assert_close(sdf2, sdf, rtol=0, atol=0)


def _compare_render(ops, scene, n_view, rgb_shrink=1.0, S_tol=1.0, precision=None):
    cache, sw, fw, ro, rd, c2w, cd, ts, te = scene
    P = cache.shape[0]
    B, Hh, Ww, _ = ro.shape
    bg = torch.ones(3)
    o32 = O.render(cache, sw, fw, ro, rd, ts, te, bg, cd, c2w, create_graph=False)
    d = torch.float64
    o64 = O.render(cache.to(d), [w.to(d) for w in sw], [w.to(d) for w in fw], ro.to(d), rd.to(d), ts.to(d), te.to(d),
                   bg.to(d), cd.to(d), c2w.to(d), create_graph=False)
    packed = ops.planes_pack(cache.cuda())
    raw = ops.render_forward_raw(packed, [w.cuda() for w in sw], [w.cuda() for w in fw], ro.reshape(-1, 3).cuda(),
                                 rd.reshape(-1, 3).cuda(), ts.cuda(), te.cuda(), Hh * Ww,
                                 ops.RenderConfig(precision=precision))
    raw = {k: v.cpu() for k, v in raw.items()}
    n_rays = B * Hh * Ww
    pairs = {
        "opacity": ("opacity", o32["opacity"].reshape(n_rays, 1), o64["opacity"].reshape(n_rays, 1)),
        "depth": ("depth", o32["depth"].reshape(n_rays, 1), o64["depth"].reshape(n_rays, 1)),
        "rgb_fg": ("comp_rgb_fg", o32["comp_rgb_fg"].reshape(n_rays, 3), o64["comp_rgb_fg"].reshape(n_rays, 3)),
        "z_variance": ("z_variance", o32["z_variance"].reshape(n_rays, 1), o64["z_variance"].reshape(n_rays, 1)),
        "weights": ("weights", o32["weights"], o64["weights"]),
        "trans": ("trans", o32["trans"], o64["trans"]),
        "sdf": ("sdf", o32["sdf"], o64["sdf"]),
        "sdf_grad": ("sdf_grad", o32["sdf_grad"], o64["sdf_grad"]),
        "features": ("features", o32["features"], o64["features"]),
    }
    report = {}
    for k, (_, w32, w64) in pairs.items():
        got = raw[k].double()
        err_hip = (got - w64).abs().max().item()
        err_cpu = (w32.double() - w64).abs().max().item()
        scale = max(w64.abs().max().item(), 1e-6)
        err_32 = (got - w32.double()).abs().max().item()  # directly against the fp32 restatement
        report[k] = (err_hip, err_cpu, scale, err_32)
        assert err_hip <= max(4 * err_cpu, 2e-5 * scale), (k, err_hip, err_cpu, scale)
        # outputs vs the fp32 reference math: rtol 1e-4 of full scale, except that the fp32 oracle itself
        # sits err_cpu away from the exact value (inv_std = 100 amplifies coordinate rounding): allow that much
        assert err_32 <= max(1e-4 * scale, 2 * err_cpu), (k, err_32, err_cpu, scale)
    from parity import report as parity_report
    parity_report("forward raw outputs (max abs): hip_vs_fp64, fp32_vs_fp64, scale, hip_vs_fp32", report)
    err_hip = (raw["normal_acc"].double() - o64["normal_acc"]).abs().max().item()
    err_cpu = (o32["normal_acc"].double() - o64["normal_acc"]).abs().max().item()
    assert err_hip <= max(4 * err_cpu, 2e-5), ("normal_acc", err_hip, err_cpu)
    return report


def test_render_fwd_small_golden(ops, golden_dir):
    k = dict(np.load(os.path.join(golden_dir, "render_small.npz")))
    sw = [T(k[f"sdf_w{i}"]) for i in range(3)]
    fw = [T(k[f"feat_w{i}"]) for i in range(3)]
    cache = T(k["cache"])
    ro, rd = T(k["rays_o"]), T(k["rays_d"])
    B, Hh, Ww, _ = ro.shape
    packed = ops.planes_pack(cache.cuda())
    raw = ops.render_forward_raw(packed, [w.cuda() for w in sw], [w.cuda() for w in fw], ro.reshape(-1, 3).cuda(),
                                 rd.reshape(-1, 3).cuda(), T(k["t_starts"]).cuda(), T(k["t_ends"]).cuda(), Hh * Ww,
                                 ops.RenderConfig())
    n_rays = B * Hh * Ww
    for name, gk in (("opacity", "opacity"), ("depth", "depth"), ("z_variance", "z_variance"), ("weights", "weights"),
                     ("sdf", "sdf"), ("sdf_grad", "sdf_grad"), ("features", "features"), ("trans", "trans")):
        want64 = T(k[f"f64_{gk}"]).reshape(raw[name].shape)
        want32 = T(k[f"f32_{gk}"]).reshape(raw[name].shape)
        err_hip = (raw[name].cpu().double() - want64).abs().max().item()
        err_cpu = (want32.double() - want64).abs().max().item()
        scale = max(want64.abs().max().item(), 1e-6)
        assert err_hip <= max(4 * err_cpu, 2e-5 * scale), (name, err_hip, err_cpu, scale)


@pytest.mark.parametrize("precision", PRECISIONS)
def test_render_fwd_c1_like(ops, precision):
    """BASELINE config[0] shape (planes 128^2, 64x64 rays, 32 samples), P=1 view=1."""
    scene = _scene(seed=5, P=1, R=128, n_view=1, Hh=64, Ww=64, S=32)
    rep = _compare_render(ops, scene, 1, precision=precision)
    print(rep)


@pytest.mark.parametrize("precision", PRECISIONS)
def test_render_fwd_multi_prompt_ragged_tile(ops, precision):
    """2 prompts x 2 views, S=45 (last tile partially filled), small planes."""
    scene = _scene(seed=6, P=2, R=32, n_view=2, Hh=5, Ww=7, S=45, near=0.4, far=2.9)
    _compare_render(ops, scene, 2, precision=precision)
