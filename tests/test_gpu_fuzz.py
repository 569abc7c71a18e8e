"""Seeded fuzz of the training render: random plane sizes, ray grids, sample counts, tilings, variance / shrink / anneal
values, stratified (non-uniform) intervals and precision modes -- forward outputs and every gradient against the fp32 /
fp64 CPU oracle at the bars of tests/parity.py (norm bars: the scenes are too small for the statistical element-wise bar)."""
import os
import random

import pytest
import torch

from oracle import cpu_ref as O

from parity import TOL_VS_FP32, check_grads, check_outputs, kink_free_rays, rel, report
from test_gpu_backward import KEYS, _hip_grads, _oracle_grads, mods  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def _case(seed):
    rnd = random.Random(1000 + seed)
    P = rnd.choice([1, 1, 2, 3])
    n_view = rnd.choice([1, 2, 3])
    R = rnd.choice([8, 12, 20, 32, 36, 64])
    Hh, Ww = rnd.randint(1, 9), rnd.randint(1, 11)
    S = rnd.choice([1, 2, 7, 16, 31, 32, 33, 50, 64])
    rc = dict(inv_std=rnd.choice([1.0, 10.0, 40.0, 100.0]), rgb_grad_shrink=rnd.choice([1.0, 0.7, 0.0]),
              cos_anneal_ratio=rnd.choice([0.0, 0.3, 1.0]))
    knobs = dict(tile_sb=rnd.choice([0, 1, 2, 4, 8, 16, 32]), tile_chunk=rnd.choice([0, 0, 1, 3, 8, 64]),
                 grad_copies=rnd.choice([1, 1, 2, 3]))
    # (the same three draws as rounds 3-4 -- exact_f32, wgrad_f32, bwd_pair -- so every seed keeps its scene: the two A/B
    # kernels of those rounds left the product library, their draws now select the fast mode)
    r_f32, r_a = rnd.random(), rnd.random()
    near, far = rnd.choice([(0.1, 4.0), (0.3, 3.2), (1.0, 2.0)])
    jittered = rnd.random() < 0.5
    r_b = rnd.random()
    knobs["precision"] = "f32" if r_f32 < 0.25 else ("split2" if (r_a < 0.15 or r_b < 0.2) else "split3")
    return P, n_view, R, Hh, Ww, S, rc, knobs, near, far, jittered


# 1 400 seeds in the suite since round 6 (round 5: 400; ~200 s on one MI355X); TT_FUZZ_SEEDS=N overrides
N_SEEDS = int(os.environ.get("TT_FUZZ_SEEDS", "1400"))
_SUITE = {}  # seed -> summary of parity.check_grads (suite gates: test_fuzz_suite_statistics below)
N_SEEDS_AUX = min(N_SEEDS, 48) // 2  # the point-query and eval fuzzes


@pytest.mark.parametrize("seed", range(N_SEEDS))
def test_random_configuration_matches_oracle(mods, seed):
    P, n_view, R, Hh, Ww, S, rck, knobs, near, far, jittered = _case(seed)
    g = torch.Generator().manual_seed(seed)
    cache = torch.randn(P, 6, 32, R, R, generator=g) * 0.5
    sw = O.init_mlp_weights([32, 64, 64, 1], g)
    fw = O.init_mlp_weights([96, 64, 64, 3], g)
    ro, rd, c2w, cd = O.make_cameras(P * n_view, Hh, Ww)
    n_rays = P * n_view * Hh * Ww
    ts, te = O.uniform_intervals(n_rays, S, near, far)
    if jittered:  # stratified edges: interval lengths differ per sample and per ray
        edges = torch.cat([ts[:, :1], te], dim=1)
        w = (far - near) / S
        edges[:, 1:-1] += (torch.rand(n_rays, S - 1, generator=g) - 0.5) * 0.9 * w
        ts, te = edges[:, :-1].contiguous(), edges[:, 1:].contiguous()
    bg = torch.rand(3, generator=g)
    proj = {n: torch.randn(P * n_view, Hh, Ww, c, generator=g) for n, c in KEYS}
    # rays with a sample on a ReLU kink of the sdf net leave the loss (parity.kink_free_rays: the normal is not defined to
    # fp32 accuracy there, one such sample moves the gradients of a 20-ray scene by 1e-3); typically 0 ... 3 % of the rays
    keep = kink_free_rays(cache, sw, fw, ro, rd, ts, te, n_view)
    # round 6: ... nor on a kink of the FEATURE network (same criterion on the texture encoding: d relu is a step, one such
    # sample moves d loss / d V1, V2 of a 200-sample scene by 1e-3 whichever side an evaluation lands on -- seed 1394 of round
    # 5's 1 400-seed run), and a masked ray leaves the EIKONAL term too (it runs over every sample's sdf_grad: seed 491)
    keep &= kink_free_rays(cache, sw, fw, ro, rd, ts, te, n_view, net="feature")
    proj = {n: v * keep.view(P * n_view, Hh, Ww, 1).to(v.dtype) for n, v in proj.items()}
    smask = keep.view(-1, 1).expand(-1, S).reshape(-1).float()
    out, l_hip, g_hip = _hip_grads(mods, cache, sw, fw, ro, rd, ts, te, bg, cd, c2w, proj, dict(rck, **knobs), sample_mask=smask)
    o32, l32, g32 = _oracle_grads(torch.float32, cache, sw, fw, ro, rd, ts, te, bg, cd, c2w, proj, rck, sample_mask=smask)
    o64, l64, g64 = _oracle_grads(torch.float64, cache, sw, fw, ro, rd, ts, te, bg, cd, c2w, proj, rck, sample_mask=smask)
    # the fp32 oracle three more times in other operation orders: their mutual distances = the order sensitivity of fp32 here
    g32a = [_oracle_grads(torch.float32, cache, sw, fw, ro, rd, ts, te, bg, cd, c2w, proj, rck, alt_order=lv, sample_mask=smask)[2] for lv in (1, 2, 3)]
    case = f"test_gpu_fuzz[{seed}] P{P} v{n_view} R{R} {Hh}x{Ww} S{S} {rck} {knobs}"
    km = keep.view(P * n_view, Hh, Ww, 1)
    masked = lambda o: {k: o[k].detach().cpu().reshape(P * n_view, Hh, Ww, -1) * km.to(o[k].dtype) for k, _ in KEYS}  # noqa: E731
    check_outputs(case, masked(out), masked(o32), masked(o64), [k for k, _ in KEYS])
    # the loss is a sum of randomly signed terms: its error is measured against their l1 mass, not against the (cancelled) sum
    mass = sum(float((o64[k].detach().double().reshape(proj[k].shape) * proj[k].double()).abs().sum()) for k in proj)
    assert abs(l_hip - l64) <= max(4 * abs(l32 - l64), 1e-5 * abs(l64), 2e-6 * mass, 1e-6), (case, l_hip, l32, l64, mass)
    nz = [i for i, t in enumerate(g64) if float(t.abs().max()) > 0]  # (rgb_grad_shrink = 0, S = 1 ...: skip all-zero grads)
    names = ["space_cache", "sdf.w1", "sdf.w2", "sdf.w3", "feat.v1", "feat.v2", "feat.v3"]
    # A case further than 1e-4 from the fp32 oracle is run again in the other two precision modes on the SAME inputs and all
    # three are recorded (information; what is ASSERTED is parity.check_grads' bar for the case's own mode).
    far = [i for i in nz if rel(g_hip[i], g32[i]) > TOL_VS_FP32]
    if far:
        twin = {names[i]: {knobs["precision"] + "_vs_fp32": rel(g_hip[i], g32[i]), "fp32_vs_fp64": rel(g32[i], g64[i]),
                           "fp32_order_sensitivity": max(rel(x[i], y[i]) for k, x in enumerate([g32] + g32a)
                                                         for y in ([g32] + g32a)[:k])} for i in far}
        for other in ("split3", "f32", "split2"):
            if other != knobs["precision"]:
                _, _, g_x = _hip_grads(mods, cache, sw, fw, ro, rd, ts, te, bg, cd, c2w, proj,
                                       dict(rck, **dict(knobs, precision=other)), sample_mask=smask)
                for i in far:
                    twin[names[i]][other + "_vs_fp32"] = rel(g_x[i], g32[i])
        report(case + " [same inputs, other precision modes]", twin)
    summ = {"mode": knobs["precision"]}
    _SUITE[seed] = summ
    try:
        check_grads(case + f" kink-free rays {int(keep.sum())}/{keep.numel()}", [g_hip[i] for i in nz], [g32[i] for i in nz],
                    [g64[i] for i in nz], names=[names[i] for i in nz], elem=False,
                    g32_alt=[[ga[i] for i in nz] for ga in g32a], fast=knobs["precision"] == "split2", summary=summ)
    except AssertionError:
        summ["failed"] = True
        raise
    for i in set(range(7)) - set(nz):
        assert float(g_hip[i].abs().max()) == 0.0, (case, names[i])


def test_fuzz_suite_statistics():
    """Suite gates of tests/parity.py (rule 3), over the seeds that ran in this process: the exception path must stay an
    exception.  (a) a seed may miss the PLAIN 1e-4 against the primary fp32 oracle only where the fp32 evaluations disagree
    among themselves (or on a kink, within 1e-4 of fp64); (b) >= SUITE_PLAIN_FRAC of all seeds meet it, p90 / p99 of the
    per-seed worst |hip - fp32| / |fp32| <= SUITE_P90 / SUITE_P99.  Written to the parity report and to
    gpurun_out/parity_summary.json (copied to profiles/ per round)."""
    import json

    import numpy as np
    from parity import ORDER_K, ROOT, SUITE_P90, SUITE_P99, SUITE_PLAIN_FRAC
    done = {s: v for s, v in _SUITE.items() if "worst_hip_vs_fp32" in v}
    if len(done) < 200:
        pytest.skip(f"suite statistics need >= 200 fuzz seeds in one process ({len(done)} ran)")
    worst = np.array([v["worst_hip_vs_fp32"] for v in done.values()])
    plain = np.array([bool(v["plain"]) for v in done.values()])
    exc = {s: v for s, v in done.items() if not v["plain"]}
    # gate (a): a seed may miss the plain bar only where the fp32 evaluations disagree among themselves (or on a kink: within
    # 1e-4 of the exact math)
    agree = {s: v for s, v in done.items() if v["max_sens"] <= TOL_VS_FP32 / ORDER_K}
    unexplained = sorted(s for s, v in agree.items() if not v["plain"] and v["worst_hip_vs_fp64"] > TOL_VS_FP32)
    by_mode = {}
    for m in ("split3", "f32", "split2"):
        w = np.array([v["worst_hip_vs_fp32"] for v in done.values() if v["mode"] == m])
        if len(w):
            by_mode[m] = {"seeds": int(len(w)), "plain_frac": float((w <= TOL_VS_FP32).mean()),
                          "p50": float(np.percentile(w, 50)), "p90": float(np.percentile(w, 90)),
                          "p99": float(np.percentile(w, 99)), "max": float(w.max())}
    summary = {"seeds": int(len(done)), "failed": sorted(s for s, v in _SUITE.items() if v.get("failed")),
               "plain_frac": float(plain.mean()), "p50": float(np.percentile(worst, 50)),
               "p90": float(np.percentile(worst, 90)), "p99": float(np.percentile(worst, 99)), "max": float(worst.max()),
               "gates": {"plain_frac_min": SUITE_PLAIN_FRAC, "p90_max": SUITE_P90, "p99_max": SUITE_P99},
               "fp32_evaluations_agree": {"seeds": int(len(agree)), "plain": int(sum(bool(v["plain"]) for v in agree.values())),
                                          "unexplained": unexplained},
               "by_mode": by_mode,
               "exceptions": {str(s): {"mode": v["mode"], "worst_hip_vs_fp32": v["worst_hip_vs_fp32"],
                                       "worst_hip_vs_fp64": v["worst_hip_vs_fp64"], "fp32_order_sensitivity": v["max_sens"],
                                       "passed_by": v["passed_by"]} for s, v in sorted(exc.items())}}
    report("test_gpu_fuzz suite statistics", summary)
    try:
        with open(os.path.join(ROOT, "gpurun_out", "parity_summary.json"), "w") as f:
            json.dump(summary, f, indent=1)
    except OSError:
        pass
    assert not unexplained, summary
    assert summary["plain_frac"] >= SUITE_PLAIN_FRAC, summary
    assert summary["p90"] <= SUITE_P90 and summary["p99"] <= SUITE_P99, summary


@pytest.mark.parametrize("seed", range(N_SEEDS_AUX))
def test_random_point_query_matches_oracle(seed):
    """geometry.forward on random point clouds (inside and outside the box), random prompt / view / plane sizes:
    outputs, d/d planes, d/d weights and d/d points (second order through sdf_grad / normal) against the oracle."""
    import triplaneturbo_amd as tt
    rnd = random.Random(5000 + seed)
    dev = torch.device("cuda", 0)
    torch.manual_seed(100 + seed)
    g = tt.find("few-step-triplane-dual-stable-diffusion")({}).to(dev)
    r_p = rnd.random()
    g.precision = "f32" if r_p < 0.25 else ("split2" if r_p < 0.45 else "split3")
    gen = torch.Generator().manual_seed(200 + seed)
    P, vpp = rnd.choice([1, 2, 3]), rnd.choice([1, 2, 4])
    N, R = rnd.choice([1, 2, 31, 32, 33, 64, 100, 257]), rnd.choice([8, 12, 32, 36, 48])
    output_normal = rnd.random() < 0.7
    spread = rnd.choice([1.8, 2.2, 3.0])
    cache = torch.randn(P, 6, 32, R, R, generator=gen) * 0.5
    pts = torch.rand(P * vpp, N, 3, generator=gen) * spread - spread / 2
    sw = [w.detach().cpu() for w in g.sdf_network.weights()]
    fw = [w.detach().cpu() for w in g.feature_network.weights()]
    keys = ("sdf", "features") + (("sdf_grad", "normal") if output_normal else ())
    proj = {k: torch.randn(P * vpp * N, 1 if k == "sdf" else 3, generator=gen) for k in keys}

    def oracle(dt, alt=False):
        x = pts.to(dt).requires_grad_(True)
        c = cache.to(dt).requires_grad_(True)
        ws = [w.to(dt).requires_grad_(True) for w in sw + fw]
        with O.alt_order(alt):
            o = O.geometry_forward(x, c.repeat_interleave(vpp, 0), ws[:3], ws[3:], output_normal=output_normal,
                                   create_graph=True)
            loss = sum((o[k] * proj[k].to(dt)).sum() for k in keys)
            return o, torch.autograd.grad(loss, [x, c] + ws)

    o32, g32 = oracle(torch.float32)
    o64, g64 = oracle(torch.float64)
    g32a = [oracle(torch.float32, alt=lv)[1] for lv in (1, 2, 3)]
    x = pts.to(dev).requires_grad_(True)
    c = cache.to(dev).requires_grad_(True)
    out = g(x, c, output_normal=output_normal)
    case = f"test_gpu_fuzz points[{seed}] P{P} vpp{vpp} N{N} R{R} normal={output_normal} precision={g.precision}"
    for k in keys:
        e_hip = (out[k].detach().cpu().double().reshape(o64[k].shape) - o64[k].detach()).abs().max().item()
        e_cpu = (o32[k].detach().double() - o64[k].detach()).abs().max().item()
        assert e_hip <= max(4 * e_cpu, 2e-5), (case, k, e_hip, e_cpu)
    loss = sum((out[k].reshape(-1, proj[k].shape[1]) * proj[k].to(dev)).sum() for k in keys)
    params = [x, c] + list(g.sdf_network.weights()) + list(g.feature_network.weights())
    g_hip = torch.autograd.grad(loss, params)
    names = ["points", "planes", "w1", "w2", "w3", "v1", "v2", "v3"]
    nz = [i for i, t in enumerate(g64) if float(t.abs().max()) > 0]
    check_grads(case, [g_hip[i].cpu().reshape(g64[i].shape) for i in nz], [g32[i] for i in nz], [g64[i] for i in nz],
                names=[names[i] for i in nz], elem=False, g32_alt=[[ga[i] for i in nz] for ga in g32a],
                fast=g.precision == "split2")


@pytest.mark.parametrize("seed", range(N_SEEDS_AUX))
def test_random_eval_render_equals_training_forward(seed):
    """The fused eval kernel with both thresholds at 0 against the training forward kernels on random configurations
    (two different work decompositions of the same arithmetic: ray tiles walked front to back vs (ray block, chunk) items)."""
    from triplaneturbo_amd import ops
    rnd = random.Random(9000 + seed)
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(300 + seed)
    P, n_view = rnd.choice([1, 2, 3]), rnd.choice([1, 2, 4])
    R = rnd.choice([8, 20, 32, 48, 64])
    Hh, Ww, S = rnd.randint(1, 20), rnd.randint(1, 20), rnd.choice([1, 2, 31, 32, 33, 64, 100, 128])
    cache = torch.randn(P, 6, 32, R, R, generator=g) * 0.5
    sw = O.init_mlp_weights([32, 64, 64, 1], g)
    fw = O.init_mlp_weights([96, 64, 64, 3], g)
    ro, rd, _, _ = O.make_cameras(P * n_view, Hh, Ww)
    ts, te = O.uniform_intervals(P * n_view * Hh * Ww, S, 0.1, 4.0)
    packed = ops.planes_pack(cache.to(dev))
    args = (packed, [w.to(dev) for w in sw], [w.to(dev) for w in fw], ro.reshape(-1, 3).to(dev),
            rd.reshape(-1, 3).to(dev), ts.to(dev), te.to(dev), Hh * Ww)
    rc = ops.RenderConfig(inv_std=rnd.choice([10.0, 100.0]), precision=rnd.choice(["split3", "split3", "f32", "split2"]),
                          cos_anneal_ratio=rnd.choice([0.0, 1.0]), tile_sb=rnd.choice([0, 1, 4]))
    image_w = rnd.choice([Ww, 0])  # 0: the kernels see no image (linear 32-ray strips)
    want = ops.render_forward_raw(*args, rc, image_w=image_w)
    got = ops.render_eval_raw(*args, rc, image_w=image_w)
    for k in ("opacity", "depth", "rgb_fg", "z_variance", "normal_acc"):
        torch.testing.assert_close(got[k], want[k], rtol=2e-5, atol=2e-5 if k == "z_variance" else 2e-6,
                                   msg=lambda m, k=k: f"{k} seed {seed} P{P} v{n_view} R{R} {Hh}x{Ww} S{S}: {m}")
