#!/usr/bin/env python
"""bench.py -- rendered rays/s (fwd+bwd) of the triplane volume-render hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without WORLD_SIZE: spawns the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W [--config 3]

One "step" = one full pass of the hot path over one batch of synthetic input, exactly what the reference's
renderer does per call in training plus the backward of a fixed scalar loss:
  tt_planes_pack -> tt_render_fwd -> renderer composite -> loss -> tt_render_bwd_geo + tt_render_bwd_tex
  -> tt_planes_unpack_grad  (-> ONE RCCL all-reduce of the flat renderer-side MLP gradient buffer when N > 1).
Workloads (weak scaling: per-GPU work is fixed as N grows):
  --config 1 (default, the headline = BASELINE.json configs[1]): per GPU 1 triplane (1,6,32,256,256) fp32 ~ 0.5*N(0,1),
             1 view of 256x256 rays, 128 uniform samples on [0.1, 4.0].
  --config 3 (BASELINE.json configs[3]: 64 prompts sharded 8-way): per GPU 8 prompts (8,6,32,256,256), one 256x256 view
             each, same samples; rank r renders prompts shard_prompts(8 N, r, N).
Inputs are resident in HBM before the timed region.  Prints ONE JSON line on rank 0.  Every number of the line can be
recomputed from profiles/ with the arithmetic in profiles/README.md.
"""
import argparse
import csv
import dataclasses
import glob
import json
import os
import shutil
import socket
import statistics
import subprocess
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# ---- algorithmic work per ray-sample --------------------------------------------
# bytes_8d: SURVEY 8(d)'s algorithmic texel bytes: forward reads 6 planes x 4 corners x 32 ch x 4 B = 3072; each
#           backward kernel ACCUMULATES the gradient of its three planes = 1536 (the re-gather of the recompute is NOT
#           algorithmic work, exactly like the recomputed FLOPs).
# macs:     algorithmic multiply-adds by the pipe that executes them in the default build (recompute and the
#           scatter-combine GEMM are not counted): "f16x3" = 2-term split-fp16 products (3 x v_mfma_f32_32x32x16_f16 per
#           32x32x16 tile), "f16x4" = the weight-gradient outer products (hi|lo pairs as adjacent k-slots: 8 fp16 MFMAs per
#           32x32x32 tile = 4 MFMA-MACs per algorithmic MAC; on the fp32 MFMA under --wgrad-f32 / --exact-f32),
#           "f32" = v_mfma_f32_32x32x2_f32, "valu" = the 64-wide output layers (plain FMAs).
ALG_PROPOSAL = {  # k_decode_rays<false,false>: the sampler's proposal pass, sdf head only (estimators.py:22-101)
    "device_kernel": "k_decode_rays", "bytes_8d": 3 * 4 * 32 * 4, "macs": {"f16x3": 2048 + 4096, "valu": 64}}
ALG = {
    "tt_render_fwd": {  # k_decode_rays<N,TEX> (+ k_march_fwd): sdf 6208 + feat 10432 + normal chain 6208 = 22848 MAC
        "device_kernel": "k_decode_rays",
        "bytes_8d": 6 * 4 * 32 * 4,
        "macs": {"f16x3": 2048 + 4096 + 6144 + 4096 + 4096 + 2048, "f32": 0, "valu": 64 + 192 + 64},
    },
    "tt_render_bwd_geo": {  # (k_march_bwd +) k_decode_bwd_geo: value chain + gradient chain, activations + weights
        "device_kernel": "k_decode_bwd_geo",
        "bytes_8d": 3 * 4 * 32 * 4,
        "macs": {"f16x3": 2048 + 4096 + 4096 + 2048 + 2048 + 4096, "f16x4": 2048 + 4096, "valu": 256},
        # executed on top of the algorithmic products: the scatter-combine GEMM (one 32-row x 32-sample x 32-channel tile per plane)
        "macs_exec_extra": {"f16x3": 3 * 1024},
    },
    "tt_render_bwd_tex": {  # k_decode_bwd_tex: activations 10432 + weight gradients 10432
        "device_kernel": "k_decode_bwd_tex",
        "bytes_8d": 3 * 4 * 32 * 4,
        "macs": {"f16x3": 4096 + 6144, "f16x4": 4096 + 6144, "valu": 384},
        # executed on top: the RECOMPUTE of k1 = V1 e, k2 = V2 k1 (mat-vec precision) and the scatter-combine GEMM
        "macs_exec_extra": {"f16x3": 6144 + 4096, "scatter_f16x3": 3 * 1024},
    },
}
ALG_POINTS = {  # --config 4: the per-point queries of the text -> mesh path (no backward)
    "tt_query_field": {  # k_query_field: sdf net 32-64-64-1 + deformation head 32-64-64-3 on the 3 geometry planes
        "device_kernel": "k_query_field", "bytes_8d": 3 * 4 * 32 * 4,
        "macs": {"f16x3": 2 * (2048 + 4096), "valu": 64 + 192}},
    "tt_query_points": {  # k_query_points<no normal, features>: sdf net + feature net 96-64-64-3 on all 6 planes
        "device_kernel": "k_query_points", "bytes_8d": 6 * 4 * 32 * 4,
        "macs": {"f16x3": 2048 + 4096 + 6144 + 4096, "valu": 64 + 192}},
}
BYTES_MARCH_FWD = 44        # t_starts, t_ends, sdf, sdf_grad(3), features(3) read; weights, trans written
BYTES_MARCH_BWD = 68        # the 9 above + trans + g_sdf_grad(3) read; (d sdf, d sdf_grad) float4 written
PEAK_F32_TFLOPS = 157.3     # dense fp32-input MFMA = fp32 vector peak (MI355X_MICROARCH.md): SURVEY 8(d)'s MLP roofline
PEAK_F16_TFLOPS = 2500.0    # dense fp16/bf16 MFMA
PEAK = {"f32": PEAK_F32_TFLOPS, "valu": PEAK_F32_TFLOPS, "f16x3": PEAK_F16_TFLOPS / 3.0, "f16x4": PEAK_F16_TFLOPS / 4.0,
        "f16x6": PEAK_F16_TFLOPS / 6.0}
PEAK_HBM_GBS = 8000.0
# the gather path: every 8-lane group of a decode wave requests one whole 128-byte texel line ("Coalesced
# gathers"), served by the CU's vector L1 (TCP) and the XCD's L2 (TCC) -- the planes are cache-resident, HBM sees a fraction.
# Peaks: L2 aggregate 34.5 TB/s (MI355X_MICROARCH.md, measured; 36.9 with L1 reuse), TCP 64 B/clk/CU x 256 CUs x 2.4 GHz
PEAK_L2_GBS = 34500.0
PEAK_TCP_GBS = 64.0 * 256 * 2.4
# texel-line bytes a kernel REQUESTS per sample: the forward-shaped kernels read every corner once (= bytes_8d); a backward
# kernel re-gathers its three planes (1536 B) and returns 1536 B of gradient lines through the atomic path
GATHER_BYTES = {"tt_render_fwd": 3072, "tt_render_bwd_geo": 1536, "tt_render_bwd_tex": 1536, "tt_decode_rays": 1536,
                "tt_query_field": 1536, "tt_query_points": 3072}
# precision modes of the MLP products (include/tt_abi.h): the arithmetic type the path computes in
DTYPES = {
    "split3": "f32 (fp32-grade: exact 3-piece fp16 operand split, 6 x v_mfma_f32_32x32x16_f16 per k-step, fp32 accumulation)",
    "f32": "f32 (fp32-input MFMA, v_mfma_f32_32x32x2_f32)",
    "split2": "f32 storage / accumulation, FAST products (2-piece fp16 split, 3 MFMA terms: ~2^-21.5 per product)",
}
DTYPE_NOTES = {
    "split3": "fp32 storage, accumulation and element-wise math; every mat-vec product of the MLP chains with both operands "
              "split EXACTLY in three fp16 pieces (hi + mid + lo = v, round-to-nearest-even, operands normalised to the top of "
              "the fp16 range) and the six product terms above 2^-33 accumulated in fp32 on the fp16 matrix pipe: product error "
              "<= 2^-24 of sum |a b| (profiles/r05_split3_probe.txt), the class of the reference's fp32 GEMM "
              "(networks.py:91-97).  Reductions over samples (weight-gradient outer products, scatter-combine GEMM) use "
              "2-piece operands with all four cross terms (DESIGN.md section 3).  `modes` below: the same build on the "
              "fp32-input MFMA and in the 2-piece fast mode",
    "f32": "TT_R_EXACT_F32: every matrix product on v_mfma_f32_32x32x2_f32",
    "split2": "TT_R_SPLIT2, the fast mode (the default of rounds 2-4): 2-piece operands, 3 x v_mfma_f32_32x32x16_f16 per "
              "k-step, ~2^-21.5 per product -- a tolerance-bounded approximation of the reference's fp32 products",
}


def kernel_roofline(name, ms, n_samples, prec, wgrad_f32=False, work=None):
    """Both forms of  max(algorithmic bytes / HBM peak, algorithmic FLOP / MFMA peak) / measured time :
      frac_8d       SURVEY 8(d) literally: bytes_8d / 8 TB/s against ALL algorithmic FLOP / 157.3 TFLOP/s (the fp32-MFMA
                    roofline 8(d) names).  A value > 1 means the kernel beats both 8(d) ceilings -- possible because the
                    planes are cache-resident (no HBM traffic per texel) and the products run on the fp16 pipe.
      frac_pipe_mix the same with every FLOP priced on the pipe that executes it (split-fp16 products: 2500/3 TFLOP/s).
    """
    a = ALG[name] if name in ALG else ALG_POINTS[name]
    macs = {"f16x3": 0, "f16x4": 0, "f16x6": 0, "f32": 0, "valu": 0, **a["macs"]}
    if prec is True or prec == "f32":  # every matrix product on the fp32 MFMA
        macs = {"f16x3": 0, "f16x4": 0, "f32": macs["f16x3"] + macs["f16x4"] + macs["f32"], "valu": macs["valu"]}
    elif prec == "split3":  # the mat-vec products carry six fp16 MFMA terms instead of three
        macs = {"f16x6": macs["f16x3"], "f16x4": macs["f16x4"], "f32": macs["f32"], "valu": macs["valu"]}
    elif wgrad_f32:  # the round-2 kernels: outer products on the fp32 MFMA
        macs = {"f16x3": macs["f16x3"], "f16x4": 0, "f32": macs["f16x4"] + macs["f32"], "valu": macs["valu"]}
    flop_per_sample = 2 * sum(macs.values())
    flop = float(flop_per_sample) * n_samples
    t_hbm = a["bytes_8d"] * n_samples / (PEAK_HBM_GBS * 1e9) * 1e3                       # ms at 8 TB/s
    t_f32 = flop / (PEAK_F32_TFLOPS * 1e12) * 1e3                                         # ms at 157.3 TFLOP/s
    t_mix = sum(2.0 * m * n_samples / (PEAK[p] * 1e12) for p, m in macs.items()) * 1e3    # ms at each pipe's peak
    extra = {}
    gb = GATHER_BYTES.get(name)
    if gb:
        inb_g = work["inbounds_plane_frac"] if work is not None else 1.0
        req = gb * n_samples * inb_g
        t_l2 = req / (PEAK_L2_GBS * 1e9) * 1e3
        live_g = work["live_tile_frac"] if work is not None else 1.0
        extra["gather"] = {
            "line_bytes_requested": int(req), "requested_GBs": round(req / (ms * 1e-3) / 1e9, 1),
            "peak_l2_GBs": PEAK_L2_GBS, "frac_l2": round(t_l2 / ms, 4), "frac_tcp": round(req / (PEAK_TCP_GBS * 1e9) * 1e3 / ms, 4),
            "note": "texel lines requested (128 B per in-bounds corner, x inbounds_plane_frac when counted) / duration against "
                    "the aggregate L2 peak (34.5 TB/s) and the L1 (TCP) peak (64 B/clk/CU): the cache path the gathers really use"}
        # the bound that can actually bind a cache-resident decode: its matrix-pipe time on the pipe it runs on, or its gather
        # path -- never above 1 (unlike frac_8d, which prices cache-served bytes against HBM and fp16-pipe FLOPs at fp32 rate)
        extra["frac_real"] = None  # filled below once t_mix is known
        extra["_t_l2"], extra["_live"] = t_l2, live_g
    if work is not None:
        # what the kernel EXECUTED (device counters, tt_render_cfg.stats): a 32-sample tile step without an in-bounds
        # texel (and, in the backward, without a non-zero upstream gradient) is skipped exactly -- its FLOPs are not
        # done; a (plane, sample) pair outside the plane touches no texel -- its bytes are not moved
        live, inb = work["live_tile_frac"], work["inbounds_plane_frac"]
        extra.update({"live_tile_frac": round(live, 4), "inbounds_plane_frac": round(inb, 4),
                 "tile_steps_visited": work["visited"], "tile_steps_executed": work["executed"],
                 "executed_tflops": round(flop * live / (ms * 1e-3) / 1e12, 2),
                 "executed_GBs": round(a["bytes_8d"] * n_samples * inb / (ms * 1e-3) / 1e9, 1),
                 "frac_8d_executed": round(max(t_hbm * inb, t_f32 * live) / ms, 4),
                 "frac_pipe_mix_executed": round(max(t_hbm * inb, t_mix * live) / ms, 4),
                 "ns_per_executed_tile_step": round(ms * 1e6 / max(work["executed"], 1), 3)})
    if "_t_l2" in extra:
        t_l2, live_g = extra.pop("_t_l2"), extra.pop("_live")
        # matrix-pipe time of what the kernel EXECUTES: the algorithmic products + the recomputed forward products and the
        # scatter-combine GEMM of the backward kernels (not algorithmic work, but real pipe time)
        t_extra = 0.0
        for pk, m in a.get("macs_exec_extra", {}).items():
            if pk == "scatter_f16x3" or (pk == "f16x3" and name == "tt_render_bwd_geo"):
                pipe = "f32" if (prec is True or prec == "f32") else "f16x3"      # the combine GEMM: two-piece operands, fp32 in f32 mode
            else:
                pipe = "f32" if (prec is True or prec == "f32") else ("f16x6" if prec == "split3" else "f16x3")
            t_extra += 2.0 * m * n_samples / (PEAK[pipe] * 1e12) * 1e3
        extra["t_pipe_executed_ms"] = round((t_mix + t_extra) * live_g, 4)
        extra["frac_real"] = round(max(t_l2, (t_mix + t_extra) * live_g) / ms, 4)
        extra["bound_real"] = "l2-gather" if t_l2 >= (t_mix + t_extra) * live_g else "matrix pipe (as executed, incl. recompute + scatter GEMM)"
    return {
        **extra,
        "avg_ms": round(ms, 4), "alg_bytes_per_sample": a["bytes_8d"], "alg_flop_per_sample": int(flop_per_sample),
        "alg_GBs": round(a["bytes_8d"] * n_samples / (ms * 1e-3) / 1e9, 1),
        "alg_tflops": round(flop / (ms * 1e-3) / 1e12, 2),
        "t_hbm_ms": round(t_hbm, 4), "t_f32mfma_ms": round(t_f32, 4), "t_pipe_mix_ms": round(t_mix, 4),
        "bound_8d": "hbm" if t_hbm >= t_f32 else "mfma", "frac_8d": round(max(t_hbm, t_f32) / ms, 4),
        "bound_pipe_mix": "hbm" if t_hbm >= t_mix else "mfma", "frac_pipe_mix": round(max(t_hbm, t_mix) / ms, 4),
    }


def make_inputs(rank, world, device, config, R=256, Hh=256, Ww=256, S=128):
    from triplaneturbo_amd import synthetic as O
    from triplaneturbo_amd.parallel import shard_prompts
    P = 1 if config == 1 else 8
    prompts = [rank] if config == 1 else list(shard_prompts(P * world, rank, world))
    caches, ros, rds, c2ws, cds = [], [], [], [], []
    for p in prompts:  # per-prompt seed: the global batch does not depend on how it is sharded
        gp = torch.Generator().manual_seed(0 + p)
        caches.append(torch.randn(1, 6, 32, R, R, generator=gp) * 0.5)
        ro, rd, c2w, cd = O.make_cameras(1, Hh, Ww, azimuth_start_deg=(90.0 if config == 1 else 45.0) * p)
        ros.append(ro), rds.append(rd), c2ws.append(c2w), cds.append(cd)
    # the MLPs are replicated parameters: identical on every rank (seed 0), as DDP guarantees
    g0 = torch.Generator().manual_seed(0)
    _ = torch.randn(1, 6, 32, R, R, generator=g0)
    sw = [w.to(device).requires_grad_(True) for w in O.init_mlp_weights([32, 64, 64, 1], g0)]
    fw = [w.to(device).requires_grad_(True) for w in O.init_mlp_weights([96, 64, 64, 3], g0)]
    g = torch.Generator().manual_seed(1000 + rank)
    cache = torch.cat(caches).to(device).requires_grad_(True)
    ts, te = O.uniform_intervals(Hh * Ww, S, 0.1, 4.0)
    proj = {k: torch.randn(P, Hh, Ww, c, generator=g).to(device) for k, c in
            (("comp_rgb", 3), ("opacity", 1), ("depth", 1), ("disparity", 1), ("comp_normal_cam_vis", 3))}
    return dict(cache=cache, sw=sw, fw=fw, ro=torch.cat(ros).to(device), rd=torch.cat(rds).to(device),
                c2w=torch.cat(c2ws).to(device), cd=torch.cat(cds).to(device), ts=ts.to(device).repeat(P, 1),
                te=te.to(device).repeat(P, 1), bg=torch.ones(3, device=device), proj=proj, prompts=prompts)


def loss_fn(out, proj, fused_eikonal=False):
    """G6 loss: seeded projections of the image-space outputs + sparsity + eikonal
    (multiprompt_dual_renderer_multistep_generator.py:635, :696-699).  fused_eikonal: the eikonal term through
    ops.eikonal_loss (tt_eikonal_fwd / _bwd: one kernel each way, same value) instead of five torch kernels each way."""
    loss = 0.0
    for k, p in proj.items():
        loss = loss + (out[k] * p).sum()
    loss = loss + (out["opacity"] ** 2 + 0.01).sqrt().mean()
    if fused_eikonal:
        from triplaneturbo_amd import ops
        loss = loss + ops.eikonal_loss(out["sdf_grad"])
    else:
        loss = loss + ((torch.linalg.norm(out["sdf_grad"], ord=2, dim=-1) - 1.0) ** 2).mean()
    return loss


def cpu_baseline(n_rays_sample=256, S=128, R=256, budget_s=25.0):
    """The CPU oracle (pure-torch restatement of the reference renderer) timed on this host: same planes,
    cameras, samples and loss as --config 1, on one image row through the middle of the object (`n_rays_sample`
    rays).  torch's intra-op threading does not scale on these gather/scatter-heavy ops (on a 256-core host 256
    threads are ~40x SLOWER than 8), so the thread count is picked by a short calibration and reported as `cores`;
    one pass at os.cpu_count() threads (SURVEY 8d's literal setting) is timed as well when it fits the budget."""
    from oracle import cpu_ref as O
    host_cores = os.cpu_count() or 1
    g = torch.Generator().manual_seed(0)
    cache = (torch.randn(1, 6, 32, R, R, generator=g) * 0.5).requires_grad_(True)
    sw = [w.requires_grad_(True) for w in O.init_mlp_weights([32, 64, 64, 1], g)]
    fw = [w.requires_grad_(True) for w in O.init_mlp_weights([96, 64, 64, 3], g)]
    ro, rd, c2w, cd = O.make_cameras(1, 256, 256)
    rows = max(1, n_rays_sample // 256)
    r0 = 128 - rows // 2
    ro, rd = ro[:, r0:r0 + rows].contiguous(), rd[:, r0:r0 + rows].contiguous()
    ts, te = O.uniform_intervals(rows * 256, S, 0.1, 4.0)
    proj = {k: torch.randn(1, rows, 256, c, generator=g) for k, c in
            (("comp_rgb", 3), ("opacity", 1), ("depth", 1), ("disparity", 1), ("comp_normal_cam_vis", 3))}

    def step():
        t0 = time.perf_counter()
        out = O.render(cache, sw, fw, ro, rd, ts, te, torch.ones(3), cd, c2w)
        loss = O.synthetic_loss(out, proj)
        torch.autograd.grad(loss, [cache] + sw + fw)
        return time.perf_counter() - t0

    t_begin = time.perf_counter()
    best_t, best_n = None, None
    for nt in sorted({n for n in (4, 8, 16, 32) if n <= host_cores} | {min(host_cores, 8)}):
        torch.set_num_threads(nt)
        step()  # warm-up at this thread count
        dt = step()
        if best_t is None or dt < best_t:
            best_t, best_n = dt, nt
        if time.perf_counter() - t_begin > budget_s * 0.5:
            break
    torch.set_num_threads(best_n)
    times = [best_t]
    while time.perf_counter() - t_begin < budget_s * 0.75 and len(times) < 5:
        times.append(step())
    dt = sorted(times)[len(times) // 2]
    all_cores = None
    if host_cores != best_n and time.perf_counter() - t_begin < budget_s * 0.8:
        torch.set_num_threads(host_cores)
        all_cores = {"cores": host_cores, "value": rows * 256 / step(), "unit": "rays/s",
                     "note": "one un-warmed pass with torch.set_num_threads(os.cpu_count()) (SURVEY 8d's literal setting)"}
        torch.set_num_threads(best_n)
    return {"value": rows * 256 / dt, "unit": "rays/s", "cores": best_n, "kind": "port",
            "sample": f"image row(s) {r0}..{r0 + rows - 1} = {rows * 256} rays (of 65536) x {S} samples, fwd+bwd of the "
                      f"same loss, fp32 torch CPU oracle, median {dt:.2f} s/pass, {best_n} threads (best of a "
                      f"calibration over 4..32; host has {host_cores} cores)",
            "all_cores": all_cores}


def march_roofline(inp, rc, ops, reps):
    """HBM roofline of the ray march: algorithmic bytes per sample x samples / HIP-event duration per launch."""
    n_rays, S = inp["ts"].shape
    ro, rd = inp["ro"].reshape(-1, 3), inp["rd"].reshape(-1, 3)
    with torch.no_grad():
        fwd = ops.render_forward_raw(ops.planes_pack(inp["cache"].detach()), [w.detach() for w in inp["sw"]],
                                     [w.detach() for w in inp["fw"]], ro, rd, inp["ts"], inp["te"],
                                     inp["ro"].shape[1] * inp["ro"].shape[2], rc, image_w=inp["ro"].shape[2])
    g_ray = {k: torch.randn_like(fwd[k]) for k in ("opacity", "depth", "rgb_fg", "normal_acc")}
    g_sdf_grad = torch.randn_like(fwd["sdf_grad"])
    ws = torch.empty((n_rays * S, 4), device=rd.device)
    t = ops.KernelTimer()
    for it in range(reps + 2):
        if it == 2:
            ops.set_kernel_timer(t)
        ops.march_forward_raw(rd, inp["ts"], inp["te"], fwd["sdf"], fwd["sdf_grad"], fwd["features"], rc, out=fwd)
        ops.march_backward_raw(rd, inp["ts"], inp["te"], fwd, fwd["sdf"], fwd["sdf_grad"], fwd["features"], rc,
                               g_opacity=g_ray["opacity"], g_depth=g_ray["depth"], g_rgb_fg=g_ray["rgb_fg"],
                               g_normal_acc=g_ray["normal_acc"], g_sdf_grad=g_sdf_grad, out=ws)
    ops.set_kernel_timer(None)
    ks = t.summary(median=True)
    ms_f, ms_b = ks["tt_march_fwd"][0], ks["tt_march_bwd"][0]
    nbytes = (BYTES_MARCH_FWD + BYTES_MARCH_BWD) * n_rays * S
    ach = nbytes / ((ms_f + ms_b) * 1e-3) / 1e9
    return {"stage": "ray march (k_march_fwd + k_march_bwd), re-timed on the live buffers", "bound": "hbm",
            "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBS, 4),
            "fwd": {"median_ms": round(ms_f, 4), "GBs": round(BYTES_MARCH_FWD * n_rays * S / (ms_f * 1e-3) / 1e9, 1)},
            "bwd": {"median_ms": round(ms_b, 4), "GBs": round(BYTES_MARCH_BWD * n_rays * S / (ms_b * 1e-3) / 1e9, 1)},
            "note": "algorithmic bytes per sample (44 fwd / 68 bwd) x samples per launch over the HIP-event duration"}


# ---- rocprofv3 --pmc passes over a short run of this same script (each pass = its own process, counters only) --------
SQ_PASS = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAVES", "SQ_ACTIVE_INST_VALU",
           "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY"]


def pmc_pass(counters, config, precision="split3", timeout_s=240):
    """One `rocprofv3 --pmc <counters> --kernel-trace` pass over `bench.py --pmc-child` (2 steps after 1 warm-up):
    returns {counter: {kernel: average value per launch}}, or (None, reason)."""
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    d = tempfile.mkdtemp(prefix="tt_pmc_", dir="/tmp")
    cmd = [exe, "--pmc"] + list(counters) + ["--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
                                            os.path.join(ROOT, "bench.py"), "--pmc-child", "--config", str(config),
                                            "--precision", precision, "--steps", "2", "--warmup", "1"]
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    try:
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        shutil.rmtree(d, ignore_errors=True)
        return None, f"rocprofv3 --pmc {' '.join(counters)} timed out"
    acc, cnt = {}, {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            c = row["Counter_Name"]
            if c not in counters:
                continue
            k = row["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
            acc.setdefault(c, {})
            cnt.setdefault(c, {})
            acc[c][k] = acc[c].get(k, 0.0) + float(row["Counter_Value"])
            cnt[c][k] = cnt[c].get(k, 0) + 1
    shutil.rmtree(d, ignore_errors=True)
    if not acc:
        return None, f"rocprofv3 --pmc {' '.join(counters)} produced no counters (rc {r.returncode}): {r.stderr[-300:]}"
    return {c: {k: acc[c][k] / cnt[c][k] for k in acc[c]} for c in acc}, None


def pmc_traffic(config, precision="split3"):
    """FETCH_SIZE and WRITE_SIZE (TCC) per kernel launch, each in its OWN pass (they do not fit one pass:
    MI355X_MICROARCH.md, rocprofv3 PMC slots).  Bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: both counters are in KiB,
    and on gfx950 FETCH_SIZE reports half of a wide coalesced read (same guide; calibrated in this workload on
    k_planes_pack: 50.3 MB read + 50.3 MB written)."""
    per = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        res, err = pmc_pass([ctr], config, precision)
        if res is None:
            return None, err
        per[ctr] = res.get(ctr, {})
    out = {}
    for k in per["FETCH_SIZE"]:
        if k in per["WRITE_SIZE"]:
            out[k] = {"fetch_KiB": round(per["FETCH_SIZE"][k], 1), "write_KiB": round(per["WRITE_SIZE"][k], 1),
                      "bytes": int((2.0 * per["FETCH_SIZE"][k] + per["WRITE_SIZE"][k]) * 1024)}
    return out, None


def pmc_pipes(config, precision="split3"):
    """What the SIMDs did, per kernel launch, from ONE pass of SQ counters (7 of the 8 SQ slots), as ratios that need no
    clock: kernel cycles = SQ_BUSY_CYCLES / 32 shader engines; 1024 SIMDs.
      mfma_util       SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024)   -- north_star's MFMA utilisation
      valu_util       SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES                -- share of a wave's life issuing VALU (incl. MFMA)
      issue_util      SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES
      wait_any        SQ_WAIT_ANY / SQ_WAVE_CYCLES                        -- wave parked on s_waitcnt / barrier
      waves_per_simd  4 SQ_WAVE_CYCLES / (kernel cycles x 1024)           -- average resident waves per SIMD"""
    res, err = pmc_pass(SQ_PASS, config, precision)
    if res is None:
        return None, err
    out = {}
    for k in res.get("SQ_BUSY_CYCLES", {}):
        g = lambda c: res.get(c, {}).get(k, 0.0)
        cyc = g("SQ_BUSY_CYCLES") / 32.0
        wc = g("SQ_WAVE_CYCLES")
        if cyc <= 0 or wc <= 0:
            continue
        out[k] = {"mfma_util": round(g("SQ_VALU_MFMA_BUSY_CYCLES") / (cyc * 1024.0), 4),
                  "valu_util": round(g("SQ_ACTIVE_INST_VALU") / wc, 4),
                  "issue_util": round(g("SQ_ACTIVE_INST_ANY") / wc, 4), "wait_any": round(g("SQ_WAIT_ANY") / wc, 4),
                  "waves_per_simd": round(4.0 * wc / (cyc * 1024.0), 3), "kernel_cycles": int(cyc),
                  "mfma_busy_cycles": int(g("SQ_VALU_MFMA_BUSY_CYCLES")), "waves": int(g("SQ_WAVES"))}
    return out, None


LDS_PASS = ["SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES"]


def pmc_lds(config, precision="split3"):
    """LDS activity per kernel launch (one more SQ pass): SQ_LDS_IDX_ACTIVE = LDS-array cycles, SQ_LDS_BANK_CONFLICT = the
    extra cycles of bank conflicts among them (MI355X_MICROARCH.md, LDS).  lds_util = LDS-array cycles / (kernel cycles x
    256 CUs) with kernel cycles = SQ_BUSY_CYCLES / 32 as in pmc_pipes: the share of the CUs' LDS cycles in use -- the
    binding resource of the forward-shaped kernels."""
    res, err = pmc_pass(LDS_PASS, config, precision)
    if res is None:
        return None, err
    out = {}
    for k in res.get("SQ_BUSY_CYCLES", {}):
        g = lambda c: res.get(c, {}).get(k, 0.0)
        cyc = g("SQ_BUSY_CYCLES") / 32.0
        if cyc <= 0:
            continue
        act = g("SQ_LDS_IDX_ACTIVE")
        out[k] = {"lds_util": round(act / (cyc * 256.0), 4), "lds_idx_active_cycles": int(act),
                  "lds_bank_conflict_cycles": int(g("SQ_LDS_BANK_CONFLICT")),
                  "bank_conflict_frac": round(g("SQ_LDS_BANK_CONFLICT") / max(act, 1.0), 4),
                  "lds_instructions": int(g("SQ_INSTS_LDS")),
                  "lds_inst_active_frac_of_wave_cycles": round(g("SQ_ACTIVE_INST_LDS") / max(g("SQ_WAVE_CYCLES"), 1.0), 4)}
    return out, None


CACHE_PASS = ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum", "TCP_TCC_READ_REQ_sum"]


def pmc_cache(config, precision="split3"):
    """What the L2 (TCC) saw per kernel launch, one more counter pass: requests, hits, misses, and the read requests the
    vector L1s (TCP) sent down.  l2_hit_rate = TCC_HIT / (TCC_HIT + TCC_MISS) (MI355X_MICROARCH.md, L2).  The byte figures
    assume the 128-byte request / line size of gfx950 and are labelled so."""
    res, err = pmc_pass(CACHE_PASS, config, precision)
    if res is None:
        return None, err
    out = {}
    for k in res.get("TCC_REQ_sum", {}):
        g = lambda c: res.get(c, {}).get(k, 0.0)
        hit, miss = g("TCC_HIT_sum"), g("TCC_MISS_sum")
        out[k] = {"tcc_requests": int(g("TCC_REQ_sum")), "tcc_hits": int(hit), "tcc_misses": int(miss),
                  "l2_hit_rate": round(hit / max(hit + miss, 1.0), 4), "tcp_tcc_read_requests": int(g("TCP_TCC_READ_REQ_sum")),
                  "l2_request_bytes_at_128B": int(g("TCC_REQ_sum") * 128)}
    return out, None


STAT_ROWS = {"tt_render_fwd": 0, "tt_render_bwd_geo": 1, "tt_render_bwd_tex": 2}


def work_fractions(stats, n_samples):
    """stats: the (3, 4) int64 tensor the decode kernels added to during ONE step (ops.RenderConfig.stats)."""
    st = stats.cpu().tolist()
    out = {}
    for name, r in STAT_ROWS.items():
        visited, executed, inb = st[r][0], st[r][1], st[r][2]
        out[name] = {"visited": visited, "executed": executed,
                     "live_tile_frac": executed / max(visited, 1), "inbounds_plane_frac": inb / (3.0 * n_samples)}
    return out


def cube_intervals(ro, rd, S, near, far, half=0.999):
    """Uniform intervals clipped per ray to the plane cube [-half, half]^3 (slab test): every sample of a ray that hits
    the cube lies inside it, so no tile step is skipped -- the `dense_scene` workload.  Rays that miss keep [near, far]."""
    o, d = ro.reshape(-1, 3), rd.reshape(-1, 3)
    inv = 1.0 / torch.where(d.abs() < 1e-9, torch.full_like(d, 1e-9), d)
    t1, t2 = (-half - o) * inv, (half - o) * inv
    tn = torch.minimum(t1, t2).max(dim=-1).values.clamp_min(near)
    tf = torch.maximum(t1, t2).min(dim=-1).values.clamp_max(far)
    hit = tf > tn + 1e-4
    tn = torch.where(hit, tn, torch.full_like(tn, near))
    tf = torch.where(hit, tf, torch.full_like(tf, far))
    edges = tn[:, None] + (tf - tn)[:, None] * torch.linspace(0.0, 1.0, S + 1, device=o.device)[None, :]
    return edges[:, :-1].contiguous(), edges[:, 1:].contiguous(), float(hit.float().mean())


def grad_error(got, ref):
    """norm-wise relative error and SURVEY 8(d)'s element-wise bar (|a-b| <= 1e-4 |b| + 1e-6 max|b|) of one gradient"""
    a, b = got.detach().double().reshape(-1), ref.detach().double().reshape(-1)
    allow = 1e-4 * b.abs() + 1e-6 * b.abs().max().clamp_min(1e-300)
    return {"rel_norm": float((a - b).norm() / b.norm().clamp_min(1e-300)),
            "elem_viol_frac": float(((a - b).abs() > allow).double().mean())}


# ---- SURVEY 8(d) secondary (reference-faithful) workloads, outside the timed region ----------------------------------
def dense_scene(device, inp, rc, steps=20, warmup=3):
    """configs[1] with every ray's samples clipped to the plane cube: no tile step can be skipped, so the kernels' cost per
    EXECUTED sample is on record next to the headline scene (where ~18 % of the tile steps are empty space)."""
    import dataclasses
    from triplaneturbo_amd import functional, ops
    S = inp["ts"].shape[1]
    ts, te, hit = cube_intervals(inp["ro"], inp["rd"], S, 0.1, 4.0)
    params = [inp["cache"]] + inp["sw"] + inp["fw"]

    def step(rcfg):
        for t in params:
            t.grad = None
        out = functional.volume_render(inp["cache"], inp["sw"], inp["fw"], inp["ro"], inp["rd"], ts, te, inp["bg"],
                                       inp["cd"], inp["c2w"], rcfg, training=True)
        loss_fn(out, inp["proj"], fused_eikonal=True).backward()

    for _ in range(warmup):
        step(rc)
    timer = ops.KernelTimer()
    ops.set_kernel_timer(timer)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(rc)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    ops.set_kernel_timer(None)
    stats = torch.zeros((3, 4), dtype=torch.int64, device=device)
    step(dataclasses.replace(rc, stats=stats))
    torch.cuda.synchronize()
    n_samples = ts.numel()
    work = work_fractions(stats, n_samples)
    kern = {k: kernel_roofline(k, v[0], n_samples, rc.prec, work=work[k])
            for k, v in timer.summary(median=True).items() if k in ALG}
    for t in params:
        t.grad = None
    return {"workload": "configs[1] planes / camera / loss, the 128 uniform samples of every ray clipped to the plane cube "
                        "(slab test, half-width 0.999): all samples in bounds on the rays that hit it",
            "rays_hitting_cube": round(hit, 4), "steps": steps, "ms_per_step": round(ms, 3),
            "rays_per_s": round(ts.shape[0] / (ms * 1e-3), 1), "kernels": kern}


def skip_workloads(device, inp, rc, train_step, train_params, steps=10):
    """The OPT-IN backward skip (tt_render_cfg.skip_eps_tex; default 0 = exact, what `value` is measured with): ms per step
    and the induced gradient error against the exact run, on the headline scene and on the training shape."""
    import dataclasses
    from triplaneturbo_amd import functional, ops
    params = [inp["cache"]] + inp["sw"] + inp["fw"]
    names = ["planes_tex", "feat.v1", "feat.v2", "feat.v3"]

    def bench_step(rcfg):
        for t in params:
            t.grad = None
        out = functional.volume_render(inp["cache"], inp["sw"], inp["fw"], inp["ro"], inp["rd"], inp["ts"], inp["te"],
                                       inp["bg"], inp["cd"], inp["c2w"], rcfg, training=True)
        loss_fn(out, inp["proj"], fused_eikonal=True).backward()
        return out

    def timed(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    def tex_grads(ps):
        return [ps[0].grad[:, 3:].clone()] + [t.grad.clone() for t in ps[4:7]]

    res = {"note": "thresholds relative to the launch's largest |cbar|_1 (cbar = d loss / d raw feature); errors of the "
                   "texture-plane / feature-net gradients against the exact (eps = 0) run of the same inputs: norm-wise and "
                   "the fraction of elements outside SURVEY 8(d)'s bar; geometry gradients are untouched"}
    out = bench_step(rc)
    sg = torch.sigmoid(out["features"].detach())
    S = inp["ts"].shape[1]
    grgb = inp["proj"]["comp_rgb"].reshape(-1, 3).repeat_interleave(S, dim=0)
    cmax = float((out["weights"].detach() * grgb * 1.002 * sg * (1 - sg)).abs().sum(-1).max())
    del sg, grgb, out
    ref = tex_grads(params)
    head = {"max_cbar": cmax, "exact_ms_per_step": round(timed(lambda: bench_step(rc)), 3)}
    for frac in (1e-5, 1e-4):
        rcs = dataclasses.replace(rc, skip_eps_tex=frac * cmax)
        ms = timed(lambda: bench_step(rcs))
        stats = torch.zeros((3, 4), dtype=torch.int64, device=device)
        bench_step(dataclasses.replace(rcs, stats=stats))
        st = stats.cpu().tolist()[2]
        head[f"eps_{frac:g}"] = {"ms_per_step": round(ms, 3), "tex_tile_steps_executed_frac": round(st[1] / max(st[0], 1), 4),
                                 "grad_error": {n: grad_error(g, r) for n, g, r in zip(names, tex_grads(params), ref)}}
    res["headline"] = head
    for t in params:
        t.grad = None
    if train_step is not None:
        rdr, run, gp, bound = train_step
        rdr.grad_skip_eps_tex = 0.0
        run()
        tg = lambda: [gp[0].grad[:, 3:].clone()] + [t.grad.clone() for t in gp[4:7]]
        ref = tg()
        tr = {"cbar_bound": bound, "exact_ms_per_step": round(timed(run), 3)}
        for frac in (1e-5, 1e-4):
            rdr.grad_skip_eps_tex = frac * bound
            ms = timed(run)
            run()
            tr[f"eps_{frac:g}"] = {"ms_per_step": round(ms, 3),
                                   "grad_error": {n: grad_error(g, r) for n, g, r in zip(names, tg(), ref)}}
        rdr.grad_skip_eps_tex = 0.0
        res["patch_renderer_training_shape"] = tr
    return res


def secondary_workloads(device, inp, steps=5, warmup=2, rc=None):
    """(i) configs[1] with the reference's sampler: 128 proposal + 64 importance samples -> 193 intervals per ray
    (proposal decode + tt_sample_importance + render + backward through the plugin);  (ii) the reference TRAINING shape:
    2 prompts x 4 views of 128x128 rays through PatchRenderer (42x42 global + 40x40 patch rays), 193 samples."""
    import triplaneturbo_amd as tt
    from triplaneturbo_amd import synthetic
    torch.manual_seed(0)
    geo = tt.find("few-step-triplane-dual-stable-diffusion")({}).to(device)
    with torch.no_grad():
        for dst, src in zip(list(geo.sdf_network.weights()) + list(geo.feature_network.weights()), inp["sw"] + inp["fw"]):
            dst.copy_(src)
    base = dict(estimator="importance", trainable_variance=False, learned_variance_init=0.4605, num_samples_per_ray=64,
                num_samples_per_ray_importance=128, near_plane=0.1, far_plane=4.0, randomized=True)
    mat, bgm = tt.find("no-material")({}), tt.find("solid-color-background")({})
    res = {}

    def timed(step):
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    # (i) importance-sampled configs[1]
    r1 = tt.find("generative-space-sdf-volume-renderer")(base, geometry=geo, material=mat, background=bgm).to(device)
    r1.train()
    r1.precision = rc.precision if rc is not None else None
    cache = inp["cache"][:1].detach().clone().requires_grad_(True)
    kw = dict(space_cache=cache, text_embed=torch.zeros(1, 77, 1024), camera_distances=inp["cd"][:1], c2w=inp["c2w"][:1])
    ro, rd, bg = inp["ro"][:1], inp["rd"][:1], inp["bg"]
    proj = {k: v[:1] for k, v in inp["proj"].items()}

    def step1():
        out = r1(ro, rd, None, bg, **kw)
        loss = loss_fn(out, proj, fused_eikonal=True)
        for p_ in [cache] + list(geo.parameters()):
            p_.grad = None
        loss.backward()

    ms = timed(step1)
    res["importance_193"] = {
        "workload": "configs[1] planes / camera / loss with the reference's sampler: 128 stratified proposal intervals "
                    "(sdf-only decode) + 64 importance samples -> 193 intervals per ray, 256x256 rays, fwd+bwd incl. the "
                    "sampling", "ms_per_step": round(ms, 3), "rays_per_s": round(65536 / (ms * 1e-3), 1),
        "samples_per_ray": 193, "proposal_decodes_per_ray": 128}
    # (ii) PatchRenderer training shape
    r2 = tt.find("patch-renderer")({"patch_size": 40, "global_downsample": 3,
                                    "base_renderer_type": "generative-space-sdf-volume-renderer", "base_renderer": base},
                                   geometry=geo, material=mat, background=bgm).to(device)
    r2.train()
    r2.base_renderer.precision = rc.precision if rc is not None else None
    P, NV = 2, 4
    gen = torch.Generator().manual_seed(1)
    cache2 = (torch.randn(P, 6, 32, 256, 256, generator=gen) * 0.5).to(device).requires_grad_(True)
    ro2, rd2, c2w2, cd2 = synthetic.make_cameras(P * NV, 128, 128)
    kw2 = dict(space_cache=cache2, text_embed=torch.zeros(P, 77, 1024), camera_distances=cd2.to(device),
               c2w=c2w2.to(device))
    ro2, rd2 = ro2.to(device), rd2.to(device)

    def step2():
        torch.manual_seed(3)  # the same random patch and jitter every step (the skip comparison needs identical samples)
        out = r2(ro2, rd2, None, bg, **kw2)
        from triplaneturbo_amd import ops
        loss = out["comp_rgb"].mean() + (out["opacity"] ** 2 + 0.01).sqrt().mean() + ops.eikonal_loss(out["sdf_grad"])
        for p_ in [cache2] + list(geo.parameters()):
            p_.grad = None
        loss.backward()

    ms = timed(step2)
    n_rays = P * NV * (42 * 42 + 40 * 40)
    res["patch_renderer_training_shape"] = {
        "workload": "reference training shape (configs/TriplaneTurbo_v1.yaml:8-9,133-150): 2 prompts x 4 views, 128x128 "
                    "rays through PatchRenderer = 42x42 global + 40x40 patch rays per view, 193 samples, planes 256^2, "
                    "fwd+bwd incl. the sampling",
        "ms_per_step": round(ms, 3), "rays_per_step": n_rays, "rays_per_s": round(n_rays / (ms * 1e-3), 1)}
    if rc is not None:
        try:
            res["dense_scene"] = dense_scene(device, inp, rc)
        except Exception as e:
            res["dense_scene"] = {"error": repr(e)}
        try:
            # |cbar|_1 <= 3 x (d mean / d comp_rgb) x max of 1.002 s (1 - s), weights <= 1
            bound = 3.0 / (P * NV * 128 * 128 * 3) * 0.2505
            res["skip"] = skip_workloads(device, inp, rc, (r2.base_renderer, step2, [cache2] + list(geo.parameters()), bound),
                                         None)
        except Exception as e:
            res["skip"] = {"error": repr(e)}
    return res


def run_config2(args, device, rank, world, dist):
    """BASELINE configs[2]: the renderer side of the distillation inner loop.  Per GPU 8 prompts x 4 views (MVDream-style),
    128 x 128 rays per view through PatchRenderer (42 x 42 global + 40 x 40 patch rays, patch_renderer.py:49-88, yaml
    :148-150), the reference's sampler (128 proposal + 64 importance samples -> 193 intervals, estimators.py:22-101), and
    per STEP the four render + backward passes of one training step (num_parts_training = 4: ...generator.py:410,500,536,
    yaml :62-63), each on its own space cache (the generator decodes new planes for every part).  Everything through the
    plugin (registry names, forward signature), fwd + bwd of a G6-style loss on the composited 128 x 128 outputs."""
    import triplaneturbo_amd as tt
    from triplaneturbo_amd import ops, synthetic
    P, NV, H, W, R, PARTS = 8, 4, 128, 128, 256, 4
    torch.manual_seed(0)
    geo = tt.find("few-step-triplane-dual-stable-diffusion")({}).to(device)
    geo.precision = args.precision
    base = dict(estimator="importance", trainable_variance=False, learned_variance_init=0.4605, num_samples_per_ray=64,
                num_samples_per_ray_importance=128, near_plane=0.1, far_plane=4.0, randomized=True)
    mat, bgm = tt.find("no-material")({}), tt.find("solid-color-background")({})
    r = tt.find("patch-renderer")({"patch_size": 40, "global_downsample": 3,
                                   "base_renderer_type": "generative-space-sdf-volume-renderer", "base_renderer": base},
                                  geometry=geo, material=mat, background=bgm).to(device)
    r.train()
    r.base_renderer.precision = args.precision
    gen = torch.Generator().manual_seed(10 + rank)
    caches = [(torch.randn(P, 6, 32, R, R, generator=gen) * 0.5).to(device).requires_grad_(True) for _ in range(PARTS)]
    ro, rd, c2w, cd = synthetic.make_cameras(P * NV, H, W)
    ro, rd, c2w, cd = ro.to(device), rd.to(device), c2w.to(device), cd.to(device)
    bg = torch.ones(3, device=device)
    proj = {k: torch.randn(P * NV, H, W, c, generator=gen).to(device) for k, c in
            (("comp_rgb", 3), ("opacity", 1), ("depth", 1), ("comp_normal_cam_vis", 3))}
    te = torch.zeros(P, 77, 1024)
    params = list(geo.parameters())

    def step():
        for p_ in params:
            p_.grad = None
        total = 0.0
        for part in range(PARTS):
            caches[part].grad = None
            out = r(ro, rd, None, bg, space_cache=caches[part], text_embed=te, camera_distances=cd, c2w=c2w)
            loss = sum((out[k] * v).sum() for k, v in proj.items()) + (out["opacity"] ** 2 + 0.01).sqrt().mean() + \
                ops.eikonal_loss(out["sdf_grad"])
            (loss / PARTS).backward()
            total = total + loss.detach()
        if world > 1:  # the renderer-side parameters' gradients: one flat all-reduce per step
            from triplaneturbo_amd.parallel import allreduce_mlp_grads
            allreduce_mlp_grads(params, dist)
        return total

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss = step()
    timer = ops.KernelTimer()
    ops.set_kernel_timer(timer)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    ops.set_kernel_timer(None)
    if world > 1:
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
    if rank != 0:
        return
    rays_render = P * NV * (42 * 42 + 40 * 40)           # rays marched per render (global + patch)
    rays_step = rays_render * PARTS
    S, S_PROP = 193, 128
    torch.cuda.synchronize()
    per_step = {}
    for label, a, b in timer.events:
        per_step[label] = per_step.get(label, 0.0) + a.elapsed_time(b) / args.steps
    n_s = rays_step * S
    kernels = {k: kernel_roofline(k, per_step[k], n_s, args.precision) for k in ALG if k in per_step}
    if "tt_decode_rays" in per_step:
        ALG["tt_decode_rays"] = ALG_PROPOSAL
        kernels["tt_decode_rays"] = kernel_roofline("tt_decode_rays", per_step["tt_decode_rays"], rays_step * S_PROP, args.precision)
    for k, v in kernels.items():
        v["launches_per_step"] = sum(1 for lab, _, _ in timer.events if lab == k) // args.steps
        v["note"] = "sum over the launches of one step (4 parts x 2 renders), samples = rays x 193 (proposal: x 128)"
    dom = max((k for k in kernels), key=lambda k: kernels[k]["avg_ms"])
    kd = kernels[dom]
    roofline = {"kernel": dom, "bound": "mfma" if kd["bound_8d"] == "mfma" else "hbm",
                "achieved": kd["alg_tflops"] if kd["bound_8d"] == "mfma" else kd["alg_GBs"],
                "peak": PEAK_F32_TFLOPS if kd["bound_8d"] == "mfma" else PEAK_HBM_GBS,
                "unit": "TFLOP/s" if kd["bound_8d"] == "mfma" else "GB/s", "frac": kd["frac_8d"],
                "frac_pipe_mix": kd["frac_pipe_mix"], "avg_kernel_ms": kd["avg_ms"], "traffic": None,
                "definition": "frac_8d over the summed launches of one step (see --config 1 for the per-launch form, the SQ / "
                              "PMC counters and the executed-work fractions)"}
    ms = dt / args.steps * 1e3
    line = {"metric": "rendered rays/sec (fwd+bwd), BASELINE configs[2]: 8 prompts x 4 views, 128x128 rays, PatchRenderer, "
                      "193 importance samples, 4 render+backward passes per step",
            "value": rays_step * world * args.steps / dt, "unit": "rays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPES[args.precision], "precision": args.precision, "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: per GPU 8 prompts x 4 views of 128x128 rays through PatchRenderer "
                                   "(42x42 global + 40x40 patch rays per view = 107 648 rays per render, 193 samples: 128 "
                                   "proposal + 64 importance), planes (8,6,32,256,256) x 4 caches, one step = the 4 render + "
                                   "backward passes of a distillation step (num_parts_training = 4)",
                       "rays_marched_per_render": rays_render, "rays_per_step": rays_step, "image_pixels_per_step": P * NV * H * W * PARTS,
                       "samples_per_ray": S, "proposal_samples_per_ray": S_PROP, "parallelism": f"dp{world}",
                       "ms_per_render_fwd_bwd": round(ms / PARTS, 3), "loss": float(loss)},
            "roofline": roofline, "kernels": kernels,
            "other_entry_points_ms_per_step": {k: round(v, 4) for k, v in per_step.items() if k not in kernels},
            "glue_ms": round(ms - sum(per_step.values()), 3), "timed_region_s": round(dt, 3)}
    if world > 1:
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has {dist.get_world_size()} ranks")
        line["multi_gpu"] = {"rccl": {"backend": dist.get_backend(), "world_size": dist.get_world_size()},
                             "note": "one flat all-reduce of the renderer-side MLP gradients per step (allreduce_mlp_grads)"}
    if world == 1 and not args.no_cpu_baseline:
        cb = cpu_baseline(S=193)
        cb["sample"] += "; stand-in for configs[2]: the same oracle pass with 193 (uniform) samples per ray -- the CPU cost per ray " \
                        "does not depend on where the samples sit"
        line["cpu_baseline"] = cb
    emit(line)


def cpu_baseline_points(n_sample=262144, budget_s=20.0):
    """--config 4: the CPU oracle's per-point decode (oracle/cpu_ref.py: geometry_forward + the deformation head through
    vanilla_mlp) timed on `n_sample` points of the 160^3 grid plus the same number of colour queries; thread count calibrated
    like cpu_baseline."""
    from oracle import cpu_ref as O
    host_cores = os.cpu_count() or 1
    g = torch.Generator().manual_seed(0)
    cache = torch.randn(1, 6, 32, 256, 256, generator=g) * 0.5
    sw = O.init_mlp_weights([32, 64, 64, 1], g)
    fw = O.init_mlp_weights([96, 64, 64, 3], g)
    dw = O.init_mlp_weights([32, 64, 64, 3], g)
    pts = torch.rand(1, n_sample, 3, generator=g) * 2 - 1

    def step():
        t0 = time.perf_counter()
        with torch.no_grad():
            out = O.geometry_forward(pts, cache, sw, fw, output_normal=False)   # sdf + colours (forward_field's sdf, export)
            O.vanilla_mlp(out["enc_geo"], dw)                                   # the deformation head
        return time.perf_counter() - t0

    t_begin = time.perf_counter()
    best_t, best_n = None, None
    for nt in sorted({n for n in (4, 8, 16, 32) if n <= host_cores} | {min(host_cores, 8)}):
        torch.set_num_threads(nt)
        step()
        dt = step()
        if best_t is None or dt < best_t:
            best_t, best_n = dt, nt
        if time.perf_counter() - t_begin > budget_s * 0.5:
            break
    torch.set_num_threads(best_n)
    times = [best_t]
    while time.perf_counter() - t_begin < budget_s * 0.8 and len(times) < 7:
        times.append(step())
    dt = sorted(times)[len(times) // 2]
    return {"value": n_sample / dt, "unit": "points/s", "cores": best_n, "kind": "port",
            "sample": f"{n_sample} uniform points of the [-1,1]^3 box (of 4 096 000 + 300 000): sdf + deformation + colour of each "
                      f"(a superset of the per-point work of the step: every point gets all three heads), fp32 torch CPU oracle, "
                      f"median {dt:.3f} s/pass, {best_n} threads (best of a calibration over 4..32; host has {host_cores} cores)"}


def run_config4(args, device):
    """BASELINE configs[4] (text -> mesh): the HIP share of the export path, everything between the generator's triplane and
    marching cubes / the OBJ writer.  One step = what `isosurface` + `colorize_mesh` ask of the geometry
    (triplaneturbo_executable/utils/mesh_exporter.py:78-105, :143-183; few_step_triplane_dual_stable_diffusion.py:375-430):
      geometry.forward_field(grid 160^3 in [-1,1]^3, space_cache)  -> sdf + deformation   (tt_planes_pack + tt_query_field)
      geometry.export(300 000 vertices, space_cache)["features"]   -> vertex colours      (tt_planes_pack + tt_query_points)
    through the plugin (registry name, isosurface_deformable_grid = True), one prompt, planes (1,6,32,256,256), N = 1 only."""
    import triplaneturbo_amd as tt
    from triplaneturbo_amd import ops
    RES, N_V, R = 160, 300_000, 256
    torch.manual_seed(0)
    geo = tt.find("few-step-triplane-dual-stable-diffusion")({"isosurface_deformable_grid": True}).to(device)
    geo.precision = args.precision
    gen = torch.Generator().manual_seed(0)
    cache = (torch.randn(1, 6, 32, R, R, generator=gen) * 0.5).to(device)
    lin = torch.linspace(-1.0, 1.0, RES, device=device)
    grid = torch.stack(torch.meshgrid(lin, lin, lin, indexing="ij"), -1).reshape(1, -1, 3).contiguous()
    # vertices of a mesh-like point set: a noisy sphere shell of radius ~0.5 (where the sdf's level set is, sphere bias 0.5)
    v = torch.nn.functional.normalize(torch.randn(N_V, 3, generator=gen), dim=-1) * (0.5 + 0.05 * torch.randn(N_V, 1, generator=gen))
    verts = v.to(device).reshape(1, N_V, 3).contiguous()

    def step():
        with torch.no_grad():
            sdf, deform = geo.forward_field(grid, cache)
            col = geo.export(verts, cache)["features"]
        return sdf, deform, col

    for _ in range(args.warmup):
        out = step()
    if args.pmc_child:
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        return
    timer = ops.KernelTimer()
    ops.set_kernel_timer(timer)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        out = step()
        b.record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ops.set_kernel_timer(None)
    sdf, deform, col = out
    assert sdf.shape == (1, RES ** 3, 1) and deform.shape == (1, RES ** 3, 3) and col.shape == (1, N_V, 3)
    n_pts = {"tt_query_field": RES ** 3, "tt_query_points": N_V}
    ksum = timer.summary(median=True)
    kernels = {k: dict(kernel_roofline(k, ms, n_pts[k], args.precision), launches=n, points_per_launch=n_pts[k],
                       points_per_s=round(n_pts[k] / (ms * 1e-3), 1))
               for k, (ms, n) in ksum.items() if k in n_pts}
    traffic, traffic_err = (None, "skipped (--no-pmc)") if args.no_pmc else pmc_traffic(4, args.precision)
    pipes, pipes_err = (None, "skipped (--no-pmc)") if args.no_pmc else pmc_pipes(4, args.precision)
    cch, cch_err = (None, "skipped (--no-pmc)") if args.no_pmc else pmc_cache(4, args.precision)
    for k, vv in kernels.items():
        dk = ALG_POINTS[k]["device_kernel"]
        if cch and dk in cch:
            vv["cache"] = dict(cch[dk], kernel=dk,
                               l2_request_GBs_at_128B=round(cch[dk]["l2_request_bytes_at_128B"] / (vv["avg_ms"] * 1e-3) / 1e9, 1))
        if traffic and dk in traffic:
            vv["pmc"] = dict(traffic[dk], kernel=dk)
        if pipes and dk in pipes:
            vv["sq"] = dict(pipes[dk], kernel=dk)
    dom = max(kernels, key=lambda k: kernels[k]["avg_ms"])
    kd = kernels[dom]
    roofline = {"kernel": dom, "bound": "mfma", "achieved": kd["alg_tflops"], "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                "frac": kd["frac_8d"], "definition": "frac_8d (SURVEY 8(d)'s fp32-MFMA roofline, as in --config 1)",
                "frac_pipe_mix": kd["frac_pipe_mix"], "frac_real": kd.get("frac_real"), "bound_real": kd.get("bound_real"),
                "gather": kd.get("gather"), "cache": kd.get("cache"), "avg_kernel_ms": kd["avg_ms"],
                "mfma_util": (kd.get("sq") or {}).get("mfma_util"), "valu_util": (kd.get("sq") or {}).get("valu_util"),
                "waves_per_simd": (kd.get("sq") or {}).get("waves_per_simd"),
                "traffic": (kd.get("pmc") or {}).get("bytes"),
                "traffic_source": ("two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of this script in this run"
                                   if traffic else f"not collected: {traffic_err}"),
                "note": "a forward-only decode over cache-resident planes: frac prices its algorithmic FLOPs at the fp32-MFMA "
                        "peak (SURVEY 8d); frac_real = max(six-term fp16-pipe time, texel-line requests / L2 peak) / duration "
                        "is the bound it can physically reach; gather = the L1/L2 path of its texel reads"}
    step_ms = sorted(a.elapsed_time(b) for a, b in ev)
    ms = dt / args.steps * 1e3
    n_total = RES ** 3 + N_V
    line = {"metric": "HIP share of text->mesh export (BASELINE configs[4]): field query 160^3 + deformation head + 300 k vertex "
                      "colours, decoded points/sec",
            "value": n_total * args.steps / dt, "unit": "points/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "ms_per_step_median_hipevent": round(statistics.median(step_ms), 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPES[args.precision], "precision": args.precision, "data": "synthetic",
            "config": {"workload": "BASELINE configs[4], HIP share: geometry.forward_field on the 160^3 isosurface grid (sdf + "
                                   "deformation head) + geometry.export vertex colours for 300 000 vertices, planes "
                                   "(1,6,32,256,256), forward only (mesh_exporter.py:78-105,143-183); marching cubes, the SD "
                                   "generator and file output are not part of the path",
                       "grid_points": RES ** 3, "vertices": N_V, "parallelism": "dp1",
                       "latency_ms": {"forward_field": round(sum(kernels[k]["avg_ms"] for k in kernels if k == "tt_query_field"), 4),
                                      "export_colours": round(sum(kernels[k]["avg_ms"] for k in kernels if k == "tt_query_points"), 4),
                                      "step_wall": round(ms, 4)},
                       "checksum": {"sdf_mean": float(sdf.mean()), "deform_abs_mean": float(deform.abs().mean()),
                                    "colour_mean": float(col.mean())}},
            "roofline": roofline, "kernels": kernels,
            "glue_ms": round(statistics.median(step_ms) - sum(v_["avg_ms"] * v_["launches"] / args.steps for v_ in kernels.values()), 4),
            "timed_region_s": round(dt, 3)}
    if not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline_points()
    emit(line)


_RESULT_FD = None


def claim_stdout():
    """stdout carries EXACTLY one JSON line.  Native libraries (gloo's `[Gloo] Rank ...`, RCCL's NCCL_DEBUG output, the HIP
    runtime) write to file descriptor 1 behind Python's back, so the descriptor itself is re-pointed: a private duplicate of
    the original stdout is kept for the result line, fd 1 becomes stderr for everything else in this process and its
    children (the spawned ranks do the same, so only rank 0's line reaches the caller's stdout)."""
    global _RESULT_FD
    if _RESULT_FD is None:
        sys.stdout.flush()
        _RESULT_FD = os.dup(1)
        os.set_inheritable(_RESULT_FD, False)
        os.dup2(2, 1)
        sys.stdout = sys.stderr


def emit(line):
    data = (json.dumps(line) + "\n").encode()
    if _RESULT_FD is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        os.write(_RESULT_FD, data)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run with N ranks on this node
    (one process per GPU, rendezvous on 127.0.0.1)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    # the ranks inherit THIS process's original stdout (claim_stdout() has not run here): each re-points its own fd 1
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 300 = a ~2.7 s timed region; --config 2: 20)")
    ap.add_argument("--warmup", type=int, default=None, help="default 10 (--config 2: 2)")
    ap.add_argument("--config", type=int, default=1, choices=(1, 2, 3, 4),
                    help="1 = BASELINE configs[1] per GPU (headline); 2 = configs[2]: the distillation loop's renders (8 prompts "
                         "x 4 views, PatchRenderer, 193 samples, 4 render+backward passes per step); 3 = configs[3]: 8 prompts x "
                         "256x256 rays per GPU; 4 = configs[4]'s HIP share: field query 160^3 + deformation head + 300 k vertex "
                         "colours (forward only, N = 1)")
    ap.add_argument("--precision", default="split3", choices=("split3", "f32", "split2"),
                    help="MLP products: split3 = fp32-grade 3-piece split on the fp16 pipe (default, the reference's "
                         "precision), f32 = fp32-input MFMA, split2 = the 2-piece fast mode of rounds 2-4")
    ap.add_argument("--exact-f32", action="store_true", help="= --precision f32")
    ap.add_argument("--torch-loss", action="store_true",
                    help="A/B: the eikonal term of the G6 loss with plain torch ops instead of ops.eikonal_loss")
    ap.add_argument("--graph", action="store_true",
                    help="A/B: capture the step in a hipGraph (torch.cuda.graph) and time replays instead of eager steps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 --pmc passes (roofline.traffic = null)")
    ap.add_argument("--no-extras", action="store_true", help="skip the exact-f32 sub-result and the secondary workloads")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)  # body of a rocprofv3 pass: steps only
    ap.add_argument("--lib-variant", default=None, help=argparse.SUPPRESS)  # dev A/B: an experiment build of the library
    ap.add_argument("--tile-sb", type=int, default=0, help=argparse.SUPPRESS)      # dev sweeps of the performance knobs
    ap.add_argument("--tile-chunk", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--grad-copies", type=int, default=1, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.steps is None:
        args.steps = {2: 20, 4: 200}.get(args.config, 300)
    if args.warmup is None:
        args.warmup = 2 if args.config == 2 else 10

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.pmc_child:
        sys.exit(spawn_ranks(args))
    claim_stdout()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and not args.pmc_child:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with --nproc-per-node {args.gpus} "
                         f"or run `python bench.py --gpus {args.gpus}` without a launcher (it spawns the ranks itself)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    n_dev = torch.cuda.device_count()
    dev_index = local_rank % n_dev  # (ranks share a GPU only in the 1-GPU smoke test of the N > 1 path)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    backend = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("TT_BENCH_BACKEND", "nccl")  # "nccl" is RCCL on ROCm; "gloo" only for smoke tests
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    from triplaneturbo_amd import _lib
    if args.lib_variant:
        _lib.use_variant(args.lib_variant)
    elif local_rank == 0:
        _lib.build()  # in-tree hipcc build; no-op when libtt_hip.so is up to date
    if world > 1:
        dist.barrier()
    from triplaneturbo_amd import functional, ops
    from triplaneturbo_amd.parallel import FlatGradBucket

    if args.exact_f32:
        args.precision = "f32"
    if args.config == 4:
        if world > 1:
            raise SystemExit("--config 4 (export latency of one prompt) is an N = 1 workload")
        run_config4(args, device)
        return
    if args.config == 2:
        run_config2(args, device, rank, world, dist)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    R, Hh, Ww, S = 256, 256, 256, 128
    inp = make_inputs(rank, world, device, args.config, R, Hh, Ww, S)
    P = inp["cache"].shape[0]
    if args.exact_f32:
        args.precision = "f32"
    rc = ops.RenderConfig(precision=args.precision, tile_sb=args.tile_sb, tile_chunk=args.tile_chunk,
                          grad_copies=args.grad_copies)
    bucket = FlatGradBucket(inp["sw"] + inp["fw"])  # MLP grads = views of one buffer: one collective, no cat / copies
    fused = not args.torch_loss

    def make_step(rcfg):
        def step():
            inp["cache"].grad = None
            bucket.zero_()
            out = functional.volume_render(inp["cache"], inp["sw"], inp["fw"], inp["ro"], inp["rd"], inp["ts"],
                                           inp["te"], inp["bg"], inp["cd"], inp["c2w"], rcfg, training=True)
            loss = loss_fn(out, inp["proj"], fused_eikonal=fused)
            loss.backward()
            bucket.all_reduce(dist)  # DDP-equivalent: one flat RCCL all-reduce on the compute stream
            return loss
        return step

    step = make_step(rc)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(0 if args.graph else args.warmup):  # (--graph warms up its own step on the capture stream)
        loss = step()
    if args.pmc_child:  # body of one rocprofv3 --pmc pass: the same steps, nothing else
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        return

    graph = None
    if args.graph:  # the whole step (incl. the all-reduce when N > 1 is NOT captured: N = 1 only) as one hipGraph
        if world > 1:
            raise SystemExit("--graph is an N = 1 A/B")
        params = [inp["cache"]] + inp["sw"] + inp["fw"]
        static_g = [torch.zeros_like(t) for t in params]

        def gstep():  # the same work with autograd.grad into static tensors (what tests/test_gpu_graph.py captures)
            out = functional.volume_render(inp["cache"], inp["sw"], inp["fw"], inp["ro"], inp["rd"], inp["ts"],
                                           inp["te"], inp["bg"], inp["cd"], inp["c2w"], rc, training=True)
            ls = loss_fn(out, inp["proj"], fused_eikonal=fused)
            for dst, gr in zip(static_g, torch.autograd.grad(ls, params)):
                dst.copy_(gr)
            return ls

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(2, args.warmup)):
                gstep()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            loss = gstep()
        run = graph.replay
    else:
        run = step

    timer = ops.KernelTimer()
    if graph is None:
        ops.set_kernel_timer(timer)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        r = run()
        if r is not None:
            loss = r
        b.record()
    barrier()
    dt_local = time.perf_counter() - t0
    ops.set_kernel_timer(None)
    dt = dt_local
    per_rank_ms, allreduce_us, ranks_seen = None, None, None
    if world > 1:
        # (gloo -- the 1-GPU smoke test of this path -- all-reduces CUDA tensors but gathers CPU tensors only)
        tmax = torch.tensor([dt_local], device=device if backend == "nccl" else "cpu", dtype=torch.float64)
        allt = [torch.zeros_like(tmax) for _ in range(world)]
        dist.all_gather(allt, tmax)
        per_rank_ms = [round(float(t) / args.steps * 1e3, 4) for t in allt]
        dt = max(float(t) for t in allt)
        # the collective alone: the flat 66.6 KB MLP-gradient all-reduce, HIP events around 50 back-to-back calls
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(5):
            bucket.all_reduce(dist)
        barrier()
        e0.record()
        for _ in range(50):
            bucket.all_reduce(dist)
        e1.record()
        torch.cuda.synchronize()
        allreduce_us = round(e0.elapsed_time(e1) / 50 * 1e3, 2)
        info = [None] * world
        props = torch.cuda.get_device_properties(dev_index)
        dist.all_gather_object(info, {"rank": rank, "local_rank": local_rank, "device": dev_index, "name": props.name,
                                      "uuid": str(getattr(props, "uuid", ""))})
        ranks_seen = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks": info,
                      "distinct_devices": len({(i["uuid"], i["device"]) for i in info})}
        if ranks_seen["world_size"] != args.gpus or len(info) != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has {ranks_seen['world_size']} ranks")
        if backend == "nccl" and ranks_seen["distinct_devices"] < world:
            # a mis-bound launch (two ranks on one GPU) must not print a scaling point
            raise SystemExit(f"bench.py: {world} ranks but only {ranks_seen['distinct_devices']} distinct GPUs: {info}")
    ms_per_step = dt / args.steps * 1e3
    n_rays = P * Hh * Ww
    value = n_rays * world * args.steps / dt
    # a window of >= 12 s of the same step (outside the K timed steps `value` comes from): several periods of a coarse
    # GPU-activity sampler (the driver's smi poll), and a check that the K-step number is sustained
    sustained = None
    if not args.pmc_child and not args.no_extras and graph is None:
        n_more = int(min(20000, max(1, 12.5 / max(dt / args.steps, 1e-4))))
        barrier()
        t1 = time.perf_counter()
        for _ in range(n_more):
            step()
        barrier()
        dts = time.perf_counter() - t1
        sustained = {"steps": n_more, "seconds": round(dts, 3), "ms_per_step": round(dts / n_more * 1e3, 4),
                     "rays_per_s_this_rank": round(n_rays * n_more / dts, 1)}

    # what the kernels executed: device-side counters of ONE more (untimed) step (every rank: the step all-reduces)
    stats = torch.zeros((3, 4), dtype=torch.int64, device=device)
    make_step(dataclasses.replace(rc, stats=stats))()
    torch.cuda.synchronize()

    if rank == 0:
        step_ms = sorted(a.elapsed_time(b) for a, b in ev)
        n_samples = n_rays * S
        if graph is not None:  # per-kernel HIP events cannot be recorded inside a replay: time eager steps for them
            ops.set_kernel_timer(timer)
            for _ in range(5):
                gstep()
            ops.set_kernel_timer(None)
        ksum = timer.summary(median=True)  # label -> (median ms, launches)
        work = work_fractions(stats, n_samples)
        kernels = {k: dict(kernel_roofline(k, ms, n_samples, args.precision, work=work.get(k)), launches=n)
                   for k, (ms, n) in ksum.items() if k in ALG}
        traffic, traffic_err = (None, "skipped (--no-pmc)") if (args.no_pmc or world > 1) else pmc_traffic(args.config, args.precision)
        pipes, pipes_err = (None, "skipped (--no-pmc)") if (args.no_pmc or world > 1) else pmc_pipes(args.config, args.precision)
        lds, lds_err = (None, "skipped (--no-pmc)") if (args.no_pmc or world > 1) else pmc_lds(args.config, args.precision)
        cch, cch_err = (None, "skipped (--no-pmc)") if (args.no_pmc or world > 1) else pmc_cache(args.config, args.precision)
        for k, v in kernels.items():
            dk = ALG[k]["device_kernel"]
            if cch and dk in cch:
                v["cache"] = dict(cch[dk], kernel=dk,
                                  l2_request_GBs_at_128B=round(cch[dk]["l2_request_bytes_at_128B"] / (v["avg_ms"] * 1e-3) / 1e9, 1))
            if traffic and dk in traffic:
                v["pmc"] = dict(traffic[dk], kernel=dk)
            if pipes and dk in pipes:
                v["sq"] = dict(pipes[dk], kernel=dk)
            if lds and dk in lds:
                v["lds"] = dict(lds[dk], kernel=dk)
        dom = max(kernels, key=lambda k: kernels[k]["avg_ms"])
        kd = kernels[dom]
        if kd["bound_8d"] == "hbm":
            roofline = {"kernel": dom, "bound": "hbm", "achieved": kd["alg_GBs"], "peak": PEAK_HBM_GBS, "unit": "GB/s"}
        else:
            roofline = {"kernel": dom, "bound": "mfma", "achieved": kd["alg_tflops"], "peak": PEAK_F32_TFLOPS,
                        "unit": "TFLOP/s"}
        roofline.update(
            frac=kd["frac_8d"], definition="frac_8d", frac_pipe_mix=kd["frac_pipe_mix"], avg_kernel_ms=kd["avg_ms"],
            live_tile_frac=kd.get("live_tile_frac"), inbounds_plane_frac=kd.get("inbounds_plane_frac"),
            frac_executed=kd.get("frac_8d_executed"), frac_pipe_mix_executed=kd.get("frac_pipe_mix_executed"),
            mfma_util=(kd.get("sq") or {}).get("mfma_util"), valu_util=(kd.get("sq") or {}).get("valu_util"),
            waves_per_simd=(kd.get("sq") or {}).get("waves_per_simd"),
            frac_real=kd.get("frac_real"), bound_real=kd.get("bound_real"), gather=kd.get("gather"), cache=kd.get("cache"),
            sq_source=("one rocprofv3 --pmc pass of SQ counters over this script in this run (bench.py: pmc_pipes); "
                       "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES / 32 x 1024 SIMDs) is the physical matrix-"
                       "pipe utilisation north_star's 40 % bar is about") if pipes else f"not collected: {pipes_err}",
            traffic=(kd.get("pmc") or {}).get("bytes"),
            traffic_source=("two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of this script in this run: "
                            "(2 x FETCH_SIZE + WRITE_SIZE) KiB per launch of " + ALG[dom]["device_kernel"])
            if traffic else f"not collected: {traffic_err}",
            note="SURVEY 8(d) literally: the dominant kernel's max(algorithmic bytes / 8 TB/s, algorithmic FLOP / 157.3 "
                 "TFLOP/s fp32-MFMA) over its median HIP-event duration on the launch stream (entry point = march + "
                 "decode kernels where fused); algorithmic bytes: 3072 B/sample forward reads, 1536 B/sample accumulated "
                 "by each backward kernel (the recompute's re-gather is not algorithmic work); frac_pipe_mix prices each "
                 "FLOP on the pipe that executes it (split-fp16: 2500/3 TFLOP/s).  frac counts ALL samples of the launch; "
                 "frac_executed = the same with the FLOPs x live_tile_frac and the bytes x inbounds_plane_frac the kernel "
                 "really executed (device counters: tile steps without an in-bounds texel are skipped exactly)")
        hbm = march_roofline(inp, rc, ops, min(args.steps, 20))
        t_all = sum(v["avg_ms"] for v in kernels.values()) * 1e-3
        stage = {  # whole-step stage rooflines over the time of ALL fused kernels
            "kernel_time_ms": round(t_all * 1e3, 4),
            "sampling_stage": {"alg_bytes_per_sample": 6144, "GBs": round(6144 * n_samples / t_all / 1e9, 1),
                               "frac_of_hbm_peak": round(6144 * n_samples / t_all / 1e9 / PEAK_HBM_GBS, 4),
                               "note": "served mostly by L1/L2/Infinity Cache (planes: 50 MB per prompt)"},
            "mlp_stage": {"alg_flop_per_sample": 137088, "tflops": round(137088.0 * n_samples / t_all / 1e12, 1),
                          "frac_of_f32_mfma_peak": round(137088.0 * n_samples / t_all / 1e12 / PEAK_F32_TFLOPS, 4)},
            "glue_ms": round(statistics.median(step_ms) - t_all * 1e3, 4),
        }
        if all("live_tile_frac" in v for v in kernels.values()) and kernels:
            ex_flop = sum(v["alg_flop_per_sample"] * v["live_tile_frac"] for v in kernels.values())
            ex_bytes = sum(v["alg_bytes_per_sample"] * v["inbounds_plane_frac"] for v in kernels.values())
            stage["sampling_stage"]["executed_frac_of_hbm_peak"] = round(ex_bytes * n_samples / t_all / 1e9 / PEAK_HBM_GBS, 4)
            stage["mlp_stage"]["executed_frac_of_f32_mfma_peak"] = round(ex_flop * n_samples / t_all / 1e12 / PEAK_F32_TFLOPS, 4)
        cfg_name = {1: "BASELINE configs[1]: per GPU 1 triplane (1,6,32,256,256), 1 view 256x256 rays",
                    3: "BASELINE configs[3]: per GPU 8 prompts (8,6,32,256,256) of a batch sharded over the GPUs, one "
                       "256x256 view each"}[args.config]
        line = {
            "metric": "rendered rays/sec (fwd+bwd) at 256x256 rays x 128 samples",
            "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "ms_per_step_median_hipevent": round(statistics.median(step_ms), 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPES[args.precision], "precision": args.precision, "dtype_note": DTYPE_NOTES[args.precision],
            "data": "synthetic",
            "config": {"workload": cfg_name + ", 128 uniform samples on [0.1,4.0], fwd + bwd of the G6 loss (d/d planes + "
                                              "d/d 6 MLP matrices, second-order normal path included)",
                       "rays_per_gpu": n_rays, "samples_per_ray": S, "prompts_per_gpu": P, "parallelism": f"dp{world}",
                       "loss": float(loss.detach()), "eikonal": "torch ops" if args.torch_loss else "ops.eikonal_loss",
                       "mode": "hipGraph replay" if graph is not None else "eager"},
            "roofline": roofline, "roofline_hbm": hbm, "kernels": kernels, "stages": stage,
            "timed_region_s": round(dt, 3), "sustained": sustained,
        }
        if world > 1:
            line["multi_gpu"] = {"per_rank_ms_per_step": per_rank_ms, "allreduce_us": allreduce_us,
                                 "allreduce_bytes": bucket.flat_grad.numel() * 4, "rccl": ranks_seen}
        if world == 1 and not args.no_extras:
            # the other two precision modes of the same build, over the SAME number of steps, with their own rooflines and
            # SQ counters (driver-visible): "f32" = the fp32-input MFMA (physical MFMA utilisation of an fp32 GEMM
            # pipe), "split2" = the fast mode
            line["modes"] = {}
            for mode in ("split3", "f32", "split2"):
                if mode == args.precision:
                    continue
                rcx = ops.RenderConfig(precision=mode)
                stepx = make_step(rcx)
                for _ in range(3):
                    stepx()
                tx = ops.KernelTimer()
                ops.set_kernel_timer(tx)
                nx = max(args.steps, 20)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(nx):
                    stepx()
                torch.cuda.synchronize()
                dtx = (time.perf_counter() - t0) / nx
                ops.set_kernel_timer(None)
                km = {k: kernel_roofline(k, ms, n_samples, mode, work=work.get(k))
                      for k, (ms, n) in tx.summary(median=True).items() if k in ALG}
                if not args.no_pmc:
                    pm, pm_err = pmc_pipes(args.config, mode)
                    for k, v in km.items():
                        dk = ALG[k]["device_kernel"]
                        if pm and dk in pm:
                            v["sq"] = dict(pm[dk], kernel=dk)
                        elif pm is None:
                            v["sq"] = {"error": pm_err}
                line["modes"][mode] = {"dtype": DTYPES[mode], "steps": nx, "ms_per_step": round(dtx * 1e3, 4),
                                       "value": round(n_rays / dtx, 1), "unit": "rays/s", "kernels": km}
            try:
                line["secondary"] = secondary_workloads(device, inp, rc=rc)
            except Exception as e:  # the headline must not depend on the extras
                line["secondary"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        emit(line)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
