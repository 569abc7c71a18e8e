#!/usr/bin/env python
"""bench.py -- rendered rays/s (fwd+bwd) of the triplane volume-render hot path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
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
Inputs are resident in HBM before the timed region.  Prints ONE JSON line on rank 0.  Roofline arithmetic:
profiles/README.md.
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# ---- algorithmic work per ray-sample --------------------------------------------
# MACs of each kernel by the pipe that executes them (recompute and the scatter-combine GEMM are NOT algorithmic work
# and are not counted): "f16x3" = 2-term split-fp16 products, 3 x v_mfma_f32_32x32x16_f16 per 32x32x16 tile;
# "f32" = v_mfma_f32_32x32x2_f32; "valu" = the 64-wide output layers (w3 / V3 rows), plain FMAs.
KERNELS = {
    "tt_render_fwd": {  # k_decode_rays<N,TEX> (+ k_march_fwd, ~0.07 ms): sdf 6208 + feat 10432 + normal chain 6208
        "macs": {"f16x3": 2048 + 4096 + 6144 + 4096 + 4096 + 2048, "f32": 0, "valu": 64 + 192 + 64},
        "bytes": 6 * 4 * 32 * 4,  # 6 planes x 4 corners x 32 ch x 4 B texel reads = 3072
    },
    "tt_render_bwd_geo": {  # (k_march_bwd, ~0.1 ms +) k_decode_bwd_geo: value chain + gradient chain, act + weights
        "macs": {"f16x3": 2048 + 4096 + 4096 + 2048 + 2048 + 4096, "f32": 2048 + 4096, "valu": 256},
        "bytes": 3 * 4 * 32 * 4 * 2,  # geometry planes: gather 1536 + gradient scatter 1536
    },
    "tt_render_bwd_tex": {  # k_decode_bwd_tex: activations 10432 + weight gradients 10432
        "macs": None,  # filled in by _tex_macs(): depends on the build (which products run on which pipe)
        "bytes": 3 * 4 * 32 * 4 * 2,
    },
}
BYTES_MARCH_FWD = 44        # t_starts, t_ends, sdf, sdf_grad(3), features(3) read; weights, trans written
BYTES_MARCH_BWD = 68        # the 9 above + trans + g_sdf_grad(3) read; (d sdf, d sdf_grad) float4 written
PEAK = {"f32": 157.3, "f16": 2500.0}        # dense MFMA TFLOP/s, MI355X_MICROARCH.md
PEAK["f16x3"] = PEAK["f16"] / 3.0           # algorithmic FLOP/s of a 3-MFMA split product at 100 % of the fp16 pipe
PEAK["valu"] = 157.3
PEAK_HBM_GBS = 8000.0


def _tex_macs(exact):
    # activations: V3^T cbar (192, valu), V2^T k2bar (4096), V1^T k1bar (6144); weights: dV3 (192, valu), dV2 (4096),
    # dV1 (6144)
    if exact:
        return {"f16x3": 0, "f32": 4096 + 6144 + 4096 + 6144, "valu": 384}
    return {"f16x3": 4096 + 6144, "f32": 4096 + 6144, "valu": 384}  # activation chain split-fp16, outer products fp32


def kernel_roofline(name, ms, n_samples, exact):
    """max(algorithmic bytes / HBM peak, algorithmic FLOP / peak of the pipe mix) / measured time."""
    k = KERNELS[name]
    macs = dict(k["macs"] or _tex_macs(exact))
    if exact:  # every matrix product on the fp32 MFMA
        macs = {"f16x3": 0, "f32": macs["f16x3"] + macs["f32"], "valu": macs["valu"]}
    flop = 2.0 * sum(macs.values()) * n_samples
    t_mfma = sum(2.0 * m * n_samples / (PEAK[p] * 1e12) for p, m in macs.items()) * 1e3  # ms at 100 % of each pipe
    t_hbm = k["bytes"] * n_samples / (PEAK_HBM_GBS * 1e9) * 1e3
    eff_peak = flop / (t_mfma * 1e-3) / 1e12
    bound = "hbm" if t_hbm >= t_mfma else "mfma"
    r = {"avg_ms": round(ms, 4), "alg_flop_per_sample": int(2 * sum(macs.values())), "alg_bytes_per_sample": k["bytes"],
         "t_mfma_ms": round(t_mfma, 4), "t_hbm_ms": round(t_hbm, 4), "mfma_peak_of_pipe_mix_tflops": round(eff_peak, 1),
         "tflops": round(flop / (ms * 1e-3) / 1e12, 2), "alg_GBs": round(k["bytes"] * n_samples / (ms * 1e-3) / 1e9, 1),
         "bound": bound, "frac": round(max(t_hbm, t_mfma) / ms, 4)}
    if r["frac"] > 1.0:
        # more algorithmic bytes per second than HBM can deliver: the texel gathers are served by L1/L2/Infinity
        # Cache (the planes are 50 MB per prompt), so HBM is not what bounds this kernel; its matrix work is
        r.update(bound="mfma", frac=round(t_mfma / ms, 4),
                 note="algorithmic texel bytes exceed the HBM roofline (cache-resident planes): priced on MFMA only")
    return r


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


def loss_fn(out, proj):
    """G6 loss: seeded projections of the image-space outputs + sparsity + eikonal
    (multiprompt_dual_renderer_multistep_generator.py:635, :696-699)."""
    loss = 0.0
    for k, p in proj.items():
        loss = loss + (out[k] * p).sum()
    loss = loss + (out["opacity"] ** 2 + 0.01).sqrt().mean()
    loss = loss + ((torch.linalg.norm(out["sdf_grad"], ord=2, dim=-1) - 1.0) ** 2).mean()
    return loss


def cpu_baseline(n_rays_sample=256, S=128, R=256, budget_s=25.0):
    """The CPU oracle (pure-torch restatement of the reference renderer) timed on this host: same planes,
    cameras, samples and loss as --config 1, on one image row through the middle of the object (`n_rays_sample`
    rays).  torch's intra-op threading does not scale on these gather/scatter-heavy ops (on a 256-core host 256
    threads are ~40x SLOWER than 8), so the thread count is picked by a short calibration and reported as `cores`."""
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
        if time.perf_counter() - t_begin > budget_s * 0.6:
            break
    torch.set_num_threads(best_n)
    times = [best_t]
    while time.perf_counter() - t_begin < budget_s and len(times) < 5:
        times.append(step())
    dt = sorted(times)[len(times) // 2]
    return {"value": rows * 256 / dt, "unit": "rays/s", "cores": best_n, "kind": "port",
            "sample": f"image row(s) {r0}..{r0 + rows - 1} = {rows * 256} rays (of 65536) x {S} samples, fwd+bwd of the "
                      f"same loss, fp32 torch CPU oracle, median {dt:.2f} s/pass, {best_n} threads (best of a "
                      f"calibration over 4..32; host has {host_cores} cores)"}


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=1, choices=(1, 3),
                    help="1 = BASELINE configs[1] per GPU (headline); 3 = configs[3]: 8 prompts x 256x256 rays per GPU")
    ap.add_argument("--exact-f32", action="store_true",
                    help="A/B: every matrix product on the fp32-input MFMA (TT_R_EXACT_F32) instead of split-fp16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    dev_index = local_rank % torch.cuda.device_count()  # (two ranks may share a GPU in the 1-GPU smoke test)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
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
    if local_rank == 0:
        _lib.build()  # in-tree hipcc build; no-op when libtt_hip.so is up to date
    if world > 1:
        dist.barrier()
    from triplaneturbo_amd import functional, ops
    from triplaneturbo_amd.parallel import FlatGradBucket

    R, Hh, Ww, S = 256, 256, 256, 128
    inp = make_inputs(rank, world, device, args.config, R, Hh, Ww, S)
    P = inp["cache"].shape[0]
    rc = ops.RenderConfig(exact_f32=args.exact_f32)
    bucket = FlatGradBucket(inp["sw"] + inp["fw"])  # MLP grads = views of one buffer: one collective, no cat / copies

    def step():
        inp["cache"].grad = None
        bucket.zero_()
        out = functional.volume_render(inp["cache"], inp["sw"], inp["fw"], inp["ro"], inp["rd"], inp["ts"], inp["te"],
                                       inp["bg"], inp["cd"], inp["c2w"], rc, training=True)
        loss = loss_fn(out, inp["proj"])
        loss.backward()
        bucket.all_reduce(dist)  # DDP-equivalent: one flat RCCL all-reduce on the compute stream
        return loss

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    timer = ops.KernelTimer()
    ops.set_kernel_timer(timer)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        loss = step()
        b.record()
    barrier()
    dt = time.perf_counter() - t0
    ops.set_kernel_timer(None)
    if world > 1:
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()
    ms_per_step = dt / args.steps * 1e3
    n_rays = P * Hh * Ww
    value = n_rays * world * args.steps / dt

    if rank == 0:
        step_ms = sorted(a.elapsed_time(b) for a, b in ev)
        ksum = timer.summary(median=True)  # label -> (median ms, launches)
        n_samples = n_rays * S
        kernels = {k: dict(kernel_roofline(k, ms, n_samples, args.exact_f32), launches=n) for k, (ms, n) in ksum.items()
                   if k in KERNELS}
        dom = max(kernels, key=lambda k: kernels[k]["avg_ms"])
        kd = kernels[dom]
        if kd["bound"] == "hbm":
            roofline = {"kernel": dom, "bound": "hbm", "achieved": kd["alg_GBs"], "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": kd["frac"]}
        else:
            roofline = {"kernel": dom, "bound": "mfma", "achieved": kd["tflops"],
                        "peak": kd["mfma_peak_of_pipe_mix_tflops"], "unit": "TFLOP/s", "frac": kd["frac"]}
        roofline.update(
            traffic=None, avg_kernel_ms=kd["avg_ms"],
            note="SURVEY 8(d): max(algorithmic bytes / 8 TB/s, algorithmic FLOP / peak of the pipe mix the kernel runs) "
                 "over the median HIP-event duration of the entry point on the launch stream; `traffic` (PMC bytes) is "
                 "not collected by this run -- see traffic_profile")
        try:  # HBM-side bytes per launch from the committed PMC passes (FETCH_SIZE x2 + WRITE_SIZE): a profile, not
            # a measurement of this run
            src = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_traffic_bytes.json"))[-1]
            tb = json.load(open(os.path.join(ROOT, "profiles", src)))
            key = {"tt_render_fwd": "k_decode_rays", "tt_render_bwd_geo": "k_decode_bwd_geo",
                   "tt_render_bwd_tex": "k_decode_bwd_tex"}[dom]
            roofline["traffic_profile"] = {"bytes_per_launch": next(v for k, v in tb.items() if key in k),
                                           "source": f"profiles/{src}"}
        except Exception:
            pass
        # the bandwidth-bound stage: the ray march (k_march_fwd / k_march_bwd), re-timed on the live buffers of one
        # more forward, outside the timed region (inside tt_render_fwd / tt_render_bwd_geo they run back to back with
        # the decode kernels, so the entry-point timers above cannot separate them)
        hbm = march_roofline(inp, rc, ops, args.steps)
        t_all = sum(v["avg_ms"] for v in kernels.values()) * 1e-3
        # SURVEY 8(d) "sampling stage": 6144 algorithmic B/sample (gather + scatter) over the time of ALL fused kernels
        hbm["fused_gather_scatter"] = {"GBs": round(6144 * n_samples / t_all / 1e9, 1),
                                       "frac_of_hbm_peak": round(6144 * n_samples / t_all / 1e9 / PEAK_HBM_GBS, 4),
                                       "note": "served mostly by L1/L2/Infinity Cache (planes: 50 MB per prompt)"}
        mlp_flop = sum(v["alg_flop_per_sample"] for v in kernels.values()) * n_samples
        cfg_name = {1: "BASELINE configs[1]: per GPU 1 triplane (1,6,32,256,256), 1 view 256x256 rays",
                    3: "BASELINE configs[3]: per GPU 8 prompts (8,6,32,256,256) of a batch sharded over the GPUs, one "
                       "256x256 view each"}[args.config]
        line = {
            "metric": "rendered rays/sec (fwd+bwd) at 256x256 rays x 128 samples",
            "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "ms_per_step_median_hipevent": round(statistics.median(step_ms), 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32",
            "dtype_note": ("TT_R_EXACT_F32: every matrix product on v_mfma_f32_32x32x2_f32" if args.exact_f32 else
                           "fp32 storage, accumulation and element-wise math; mat-vec products as 2-term split-fp16 "
                           "(hi + lo/2048, 22-bit significands, 3 fp16 MFMAs, fp32 accumulate) except where `kernels` "
                           "lists f32 MACs; --exact-f32 runs everything on the fp32 MFMA"),
            "data": "synthetic",
            "config": {"workload": cfg_name + ", 128 uniform samples on [0.1,4.0], fwd + bwd of the G6 loss (d/d planes + "
                                              "d/d 6 MLP matrices, second-order normal path included)",
                       "rays_per_gpu": n_rays, "samples_per_ray": S, "prompts_per_gpu": P, "parallelism": f"dp{world}",
                       "loss": float(loss.detach())},
            "roofline": roofline, "roofline_hbm": hbm, "kernels": kernels,
            "mlp_stage": {"alg_tflop_per_step": round(mlp_flop / 1e12, 4),
                          "tflops_over_kernel_time": round(mlp_flop / t_all / 1e12, 1)},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
