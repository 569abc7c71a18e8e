#!/usr/bin/env python
"""bench.py -- rendered rays/s (fwd+bwd) of the triplane volume-render hot path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one full pass of the hot path over one batch of synthetic input, exactly what the reference's
renderer does per call in training plus the backward of a fixed scalar loss:
  tt_planes_pack -> tt_render_fwd -> renderer composite -> loss -> tt_render_bwd_geo + tt_render_bwd_tex
  -> tt_planes_unpack_grad  (-> RCCL all-reduce of the renderer-side MLP grads when N > 1).
Workload at every N (weak scaling, one prompt per GPU):  BASELINE.json configs[1]
  planes (1,6,32,256,256) fp32 ~ 0.5*N(0,1), 1 view of 256x256 rays, 128 uniform samples on [0.1, 4.0].
Inputs are resident in HBM before the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic work per ray-sample (DESIGN.md "Measurement")
FLOP_FWD = 2 * 22848        # sdf 6208 + feat 10432 + normal chain 6208 MAC
FLOP_BWD_GEO = 2 * 24832    # sdf value chain + gradient chain, weights + activations
FLOP_BWD_TEX = 2 * 20864    # feature net, weights + activations
BYTES_FWD = 3072            # 6 planes x 4 corners x 32 ch x 4 B texel reads
BYTES_BWD = 3072            # same footprint of plane-gradient accumulation
BYTES_MARCH_FWD = 44        # t_starts, t_ends, sdf, sdf_grad(3), features(3) read; weights, trans written
BYTES_MARCH_BWD = 68        # the 9 above + trans + g_sdf_grad(3) read; (d sdf, d sdf_grad) float4 written
PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0


def make_inputs(rank, device, R=256, Hh=256, Ww=256, S=128):
    from triplaneturbo_amd import synthetic as O
    g = torch.Generator().manual_seed(0 + rank)
    cache = (torch.randn(1, 6, 32, R, R, generator=g) * 0.5).to(device).requires_grad_(True)
    sw = [w.to(device).requires_grad_(True) for w in O.init_mlp_weights([32, 64, 64, 1], g)]
    fw = [w.to(device).requires_grad_(True) for w in O.init_mlp_weights([96, 64, 64, 3], g)]
    ro, rd, c2w, cd = O.make_cameras(1, Hh, Ww, azimuth_start_deg=90.0 * rank)
    ts, te = O.uniform_intervals(Hh * Ww, S, 0.1, 4.0)
    proj = {k: torch.randn(1, Hh, Ww, c, generator=g).to(device) for k, c in
            (("comp_rgb", 3), ("opacity", 1), ("depth", 1), ("disparity", 1), ("comp_normal_cam_vis", 3))}
    return dict(cache=cache, sw=sw, fw=fw, ro=ro.to(device), rd=rd.to(device), c2w=c2w.to(device), cd=cd.to(device),
                ts=ts.to(device), te=te.to(device), bg=torch.ones(3, device=device), proj=proj)


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
    cameras, samples and loss, on the first `n_rays_sample` rays of the workload (one image row).  torch's
    intra-op threading does not scale on these gather/scatter-heavy ops (on a 256-core host 256 threads are ~40x
    SLOWER than 8), so the thread count is picked by a short calibration and reported as `cores`."""
    from oracle import cpu_ref as O
    host_cores = os.cpu_count() or 1
    g = torch.Generator().manual_seed(0)
    cache = (torch.randn(1, 6, 32, R, R, generator=g) * 0.5).requires_grad_(True)
    sw = [w.requires_grad_(True) for w in O.init_mlp_weights([32, 64, 64, 1], g)]
    fw = [w.requires_grad_(True) for w in O.init_mlp_weights([96, 64, 64, 3], g)]
    ro, rd, c2w, cd = O.make_cameras(1, 256, 256)
    rows = max(1, n_rays_sample // 256)
    ro, rd = ro[:, :rows].contiguous(), rd[:, :rows].contiguous()
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
            "sample": f"first {rows * 256} rays (of 65536) x {S} samples, fwd+bwd of the same loss, fp32 torch CPU "
                      f"oracle, {dt:.2f} s/pass, {best_n} threads (best of a calibration; host has {host_cores} cores)"}


def march_roofline(inp, rc, ops, reps):
    """HBM roofline of the ray march: algorithmic bytes per sample x samples / HIP-event duration per launch."""
    n_rays, S = inp["ts"].shape
    ro, rd = inp["ro"].reshape(-1, 3), inp["rd"].reshape(-1, 3)
    with torch.no_grad():
        fwd = ops.render_forward_raw(ops.planes_pack(inp["cache"].detach()), [w.detach() for w in inp["sw"]],
                                     [w.detach() for w in inp["fw"]], ro, rd, inp["ts"], inp["te"], n_rays, rc,
                                     image_w=inp["ro"].shape[2])
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
    ks = t.summary()
    ms_f, ms_b = ks["tt_march_fwd"][0], ks["tt_march_bwd"][0]
    nbytes = (BYTES_MARCH_FWD + BYTES_MARCH_BWD) * n_rays * S
    ach = nbytes / ((ms_f + ms_b) * 1e-3) / 1e9
    return {"stage": "ray march (k_march_fwd + k_march_bwd)", "bound": "hbm", "achieved": round(ach, 1),
            "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBS, 4),
            "fwd": {"avg_ms": round(ms_f, 4), "GBs": round(BYTES_MARCH_FWD * n_rays * S / (ms_f * 1e-3) / 1e9, 1)},
            "bwd": {"avg_ms": round(ms_b, 4), "GBs": round(BYTES_MARCH_BWD * n_rays * S / (ms_b * 1e-3) / 1e9, 1)},
            "note": "algorithmic bytes per sample (44 fwd / 68 bwd) x samples per launch over the HIP-event duration"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
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
    from triplaneturbo_amd.parallel import allreduce_mlp_grads

    R, Hh, Ww, S = 256, 256, 256, 128
    inp = make_inputs(rank, device, R, Hh, Ww, S)
    rc = ops.RenderConfig()
    params = [inp["cache"]] + inp["sw"] + inp["fw"]

    def step():
        for t in params:
            t.grad = None
        out = functional.volume_render(inp["cache"], inp["sw"], inp["fw"], inp["ro"], inp["rd"], inp["ts"], inp["te"],
                                       inp["bg"], inp["cd"], inp["c2w"], rc, training=True)
        loss = loss_fn(out, inp["proj"])
        loss.backward()
        if world > 1:
            allreduce_mlp_grads(inp["sw"] + inp["fw"], dist)  # DDP-equivalent: one flat RCCL all-reduce
        return loss

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
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
        dt = tmax.item()
    ms_per_step = dt / args.steps * 1e3
    n_rays = Hh * Ww
    value = n_rays * world * args.steps / dt

    if rank == 0:
        ksum = timer.summary()  # label -> (avg ms, launches)
        n_samples = n_rays * S
        flops = {"tt_render_fwd": FLOP_FWD, "tt_render_bwd_geo": FLOP_BWD_GEO, "tt_render_bwd_tex": FLOP_BWD_TEX}
        kernels = {}
        for k, (ms, n) in ksum.items():
            kernels[k] = {"avg_ms": round(ms, 4), "launches": n,
                          "tflops": round(flops[k] * n_samples / (ms * 1e-3) / 1e12, 3)}
        dom = max(ksum, key=lambda k: ksum[k][0])
        ach = flops[dom] * n_samples / (ksum[dom][0] * 1e-3) / 1e12
        traffic = None  # HBM-side bytes per launch from the committed PMC passes (FETCH_SIZE x2 + WRITE_SIZE)
        try:
            tb = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic_bytes.json")))
            key = {"tt_render_fwd": "k_decode_rays", "tt_render_bwd_geo": "k_decode_bwd_geo",
                   "tt_render_bwd_tex": "k_decode_bwd_tex"}[dom]
            traffic = next(v for k, v in tb.items() if key in k)
        except Exception:
            pass
        roofline = {"kernel": dom, "bound": "mfma", "achieved": round(ach, 3), "peak": PEAK_F32_MFMA_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic,
                    "avg_kernel_ms": round(ksum[dom][0], 4),
                    "note": "algorithmic FLOP/sample x samples per launch over the HIP-event duration, against the "
                            "fp32-input MFMA peak (v_mfma_f32_32x32x2_f32), which is what most of the texture "
                            "backward still uses; the forward and geometry-backward mat-vec chains run as 2-term split-fp16 MFMAs "
                            "(22-bit products, fp32 accumulate, 3 MFMAs at 16x the fp32 rate), so their `tflops` in "
                            "`kernels` may exceed this peak"}
        # the bandwidth-bound stage: the ray march (k_march_fwd / k_march_bwd), re-timed on the live buffers of one
        # more forward, outside the timed region (inside tt_render_fwd / tt_render_bwd_geo they run back to back with
        # the decode kernels, so the entry-point timers above cannot separate them)
        hbm = march_roofline(inp, rc, ops, args.steps)
        # texel traffic of the fused decode kernels, for scale: served by L1/L2/MALL, not an HBM-roofline claim
        t_all = sum(v[0] for v in ksum.values()) * 1e-3
        hbm["fused_gather_scatter_GBs"] = round((BYTES_FWD + BYTES_BWD) * n_samples / t_all / 1e9, 1)
        line = {
            "metric": "rendered rays/sec (fwd+bwd) at 256x256 rays x 128 samples",
            "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "dtype_note": "fp32 storage, accumulation and element-wise math; matrix products of the "
            "forward decode, the geometry backward and the first layer of the texture backward as hi + lo/2048 fp16 "
            "splits (22 significand bits)",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: per GPU 1 triplane (1,6,32,256,256), 1 view 256x256 rays, "
                                   "128 uniform samples on [0.1,4.0], fwd + bwd of the G6 loss "
                                   "(d/d planes + d/d 6 MLP matrices, second-order normal path included)",
                       "rays_per_gpu": n_rays, "samples_per_ray": S, "parallelism": f"dp{world}",
                       "loss": float(loss.detach())},
            "roofline": roofline, "roofline_hbm": hbm, "kernels": kernels,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
