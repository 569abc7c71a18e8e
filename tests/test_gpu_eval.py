"""Eval-mode render on the fused decode + march kernel (tt_render_eval): with both thresholds at 0 it must reproduce the
per-ray outputs of the training forward (tt_render_fwd) and the oracle; with early termination ON the induced error must
stay inside the documented bound while the number of decoded tile steps drops (north_star: "wavefront ballot ... for
ray compaction and early termination"; SURVEY 8d: "early-termination OFF for parity, ON for throughput with the induced
error reported")."""
import time

import pytest
import torch

from oracle import cpu_ref as O

from parity import PRECISIONS, report

pytestmark = pytest.mark.gpu


def _setup(P, R, n_view, Hh, Ww, S, seed, near=0.3, far=3.2):
    from triplaneturbo_amd import ops
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(seed)
    cache = torch.randn(P, 6, 32, R, R, generator=g) * 0.5
    sw = O.init_mlp_weights([32, 64, 64, 1], g)
    fw = O.init_mlp_weights([96, 64, 64, 3], g)
    ro, rd, c2w, cd = O.make_cameras(P * n_view, Hh, Ww)
    ts, te = O.uniform_intervals(P * n_view * Hh * Ww, S, near, far)
    packed = ops.planes_pack(cache.to(dev))
    args = (packed, [w.to(dev) for w in sw], [w.to(dev) for w in fw], ro.reshape(-1, 3).to(dev),
            rd.reshape(-1, 3).to(dev), ts.to(dev), te.to(dev), Hh * Ww)
    return ops, args, (cache, sw, fw, ro, rd, ts, te, c2w, cd)


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("P,R,n_view,Hh,Ww,S,seed", [(1, 32, 1, 8, 8, 32, 1), (2, 48, 2, 13, 9, 45, 2),
                                                     (1, 64, 1, 33, 30, 70, 3)])
def test_eval_kernel_matches_training_forward_and_oracle(P, R, n_view, Hh, Ww, S, seed, precision):
    ops, args, (cache, sw, fw, ro, rd, ts, te, c2w, cd) = _setup(P, R, n_view, Hh, Ww, S, seed)
    rc = ops.RenderConfig(inv_std=100.0, precision=precision)
    want = ops.render_forward_raw(*args, rc, image_w=Ww)
    got = ops.render_eval_raw(*args, rc, image_w=Ww)  # thresholds 0: nothing approximated
    for k in ("opacity", "depth", "rgb_fg", "z_variance", "normal_acc"):
        # (z_variance: one-pass sum w t^2 - 2 D^2 + D^2 sum w here vs two passes there: ~1e-7 of t^2 <= 16)
        torch.testing.assert_close(got[k], want[k], rtol=2e-5, atol=1e-5 if k == "z_variance" else 2e-6)
    d = torch.float64
    o64 = O.render(cache.to(d), [w.to(d) for w in sw], [w.to(d) for w in fw], ro.to(d), rd.to(d), ts.to(d), te.to(d),
                   torch.ones(3, dtype=d), cd.to(d), c2w.to(d), create_graph=False, training=False)
    o32 = O.render(cache, sw, fw, ro, rd, ts, te, torch.ones(3), cd, c2w, create_graph=False, training=False)
    n = P * n_view * Hh * Ww
    for k, ok in (("opacity", "opacity"), ("depth", "depth"), ("rgb_fg", "comp_rgb_fg"), ("z_variance", "z_variance")):
        w64 = o64[ok].reshape(n, -1)
        e_hip = (got[k].cpu().double() - w64).abs().max().item()
        e_cpu = (o32[ok].reshape(n, -1).double() - w64).abs().max().item()
        assert e_hip <= max(4 * e_cpu, 2e-5), (k, e_hip, e_cpu)


def test_early_termination_error_bound_and_savings():
    P, R, n_view, Hh, Ww, S = 1, 128, 2, 64, 64, 128
    ops, args, _ = _setup(P, R, n_view, Hh, Ww, S, 7, near=0.1, far=4.0)
    rc = ops.RenderConfig(inv_std=100.0)
    exact = ops.render_eval_raw(*args, rc, image_w=Ww)
    st0 = torch.zeros(2, dtype=torch.int64, device="cuda")
    ops.render_eval_raw(*args, rc, image_w=Ww, stats=st0)
    rows = {}
    for eps in (1e-4, 1e-3):
        st = torch.zeros(2, dtype=torch.int64, device="cuda")
        got = ops.render_eval_raw(*args, rc, image_w=Ww, transmittance_eps=eps, weight_eps=eps / S, stats=st)
        err = {k: (got[k] - exact[k]).abs().max().item() for k in ("opacity", "rgb_fg", "depth", "normal_acc")}
        # dropped weights of a ray sum to < eps (transmittance) + S * eps / S (skipped colours); depth carries t <= far
        assert err["opacity"] <= 1.01 * eps and err["rgb_fg"] <= 2.05 * eps and err["depth"] <= 4.0 * eps * 1.01, err
        assert err["normal_acc"] <= 1.01 * eps
        rows[f"eps={eps:g}"] = dict(err, geo_steps=int(st[0]), tex_steps=int(st[1]), geo_steps_exact=int(st0[0]),
                                    tex_steps_exact=int(st0[1]))
        assert st[0] < st0[0] and st[1] < st0[1]  # rays behind the surface stop; empty space skips the texture net
    report("eval early termination (2 views 64x64 x 128 samples, planes 128^2): max abs error and decoded tile steps", rows)
    print(rows)


def test_eval_full_size_timing_and_plugin_path():
    """BASELINE configs[1] shape in eval mode through the plugin: fused kernel, per-ray outputs only."""
    import bench
    import triplaneturbo_amd as tt
    dev = torch.device("cuda", 0)
    inp = bench.make_inputs(0, 1, dev, 1)
    torch.manual_seed(0)
    g = tt.find("few-step-triplane-dual-stable-diffusion")({}).to(dev)
    r = tt.find("generative-space-sdf-volume-renderer")(
        dict(estimator="importance", trainable_variance=False, learned_variance_init=0.4605, num_samples_per_ray=64,
             num_samples_per_ray_importance=128, near_plane=0.1, far_plane=4.0), geometry=g,
        material=tt.find("no-material")({}), background=tt.find("solid-color-background")({})).to(dev)
    r.eval()
    kw = dict(space_cache=inp["cache"].detach(), text_embed=torch.zeros(1, 77, 1024), camera_distances=inp["cd"],
              c2w=inp["c2w"], t_starts=inp["ts"], t_ends=inp["te"])
    res = {}
    for eps in (0.0, 1e-4):
        r.eval_termination_eps = eps
        with torch.no_grad():
            out = r(inp["ro"], inp["rd"], None, torch.ones(3, device=dev), **kw)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                out = r(inp["ro"], inp["rd"], None, torch.ones(3, device=dev), **kw)
            torch.cuda.synchronize()
        res[eps] = (out, (time.perf_counter() - t0) / 5 * 1e3)
        assert "weights" not in out and out["comp_rgb"].shape == (1, 256, 256, 3)
    err = (res[1e-4][0]["comp_rgb"] - res[0.0][0]["comp_rgb"]).abs().max().item()
    assert err <= 2.05e-4, err
    report("eval render 256x256 x 128 samples through the plugin (ms per call; max |d comp_rgb|)",
           {"eps=0": res[0.0][1], "eps=1e-4": res[1e-4][1], "comp_rgb_err": err})
    print({"ms eps=0": res[0.0][1], "ms eps=1e-4": res[1e-4][1], "err": err})
