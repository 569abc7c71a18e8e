"""Dev tool (tuning build): where do the cycles of k_decode_bwd_tex go?  s_memtime stamps between the phases of a tile
step, summed over all waves of the bench workload.  usage: python tools/phase_cycles.py [training]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from triplaneturbo_amd import _lib  # noqa: E402

if os.environ.get("TT_LIB_VARIANT"):  # a -DTT_TUNING experiment build (tools/build_variants.py with TT_VARIANT_TUNING=1)
    _lib.use_variant(os.environ["TT_LIB_VARIANT"])
else:
    _lib.use_tuning_build()
import bench  # noqa: E402
from triplaneturbo_amd import functional, ops  # noqa: E402

dev = torch.device("cuda", 0)
if len(sys.argv) > 1 and sys.argv[1] == "training":
    # the reference's training shapes: 2 prompts x 4 views, PatchRenderer 42^2 + 40^2 rays, 128 + 64 importance samples
    import triplaneturbo_amd as tt
    from triplaneturbo_amd import synthetic
    torch.manual_seed(0)
    geo = tt.find("few-step-triplane-dual-stable-diffusion")({}).to(dev)
    base = dict(estimator="importance", trainable_variance=False, learned_variance_init=0.4605, num_samples_per_ray=64,
                num_samples_per_ray_importance=128, near_plane=0.1, far_plane=4.0, randomized=True)
    rend = tt.find("patch-renderer")({"patch_size": 40, "global_downsample": 3,
                                      "base_renderer_type": "generative-space-sdf-volume-renderer",
                                      "base_renderer": base}, geometry=geo, material=tt.find("no-material")({}),
                                     background=tt.find("solid-color-background")({})).to(dev)
    rend.train()
    gen = torch.Generator().manual_seed(1)
    cache = (torch.randn(2, 6, 32, 256, 256, generator=gen) * 0.5).to(dev).requires_grad_(True)
    ro, rd, c2w, cd = synthetic.make_cameras(8, 128, 128)
    kw = dict(space_cache=cache, text_embed=torch.zeros(2, 77, 1024), camera_distances=cd.to(dev), c2w=c2w.to(dev))
    ro, rd, bgc = ro.to(dev), rd.to(dev), torch.ones(3, device=dev)

    # TT_LOSS_KEYS / TT_LOSS_SCALE: seeded projections of those outputs (bench.py --config 2's loss) instead of comp_rgb.mean()
    _keys = [k for k in os.environ.get("TT_LOSS_KEYS", "").split(",") if k]
    _scale = float(os.environ.get("TT_LOSS_SCALE", "1"))
    _gen = torch.Generator().manual_seed(5)
    _proj = {k: torch.randn(8, 128, 128, {"comp_rgb": 3, "opacity": 1, "depth": 1, "comp_normal_cam_vis": 3}[k],
                            generator=_gen).to(dev) * _scale for k in _keys}

    def step():
        out = rend(ro, rd, None, bgc, **kw)
        loss = (sum((out[k] * v).sum() for k, v in _proj.items()) if _keys else out["comp_rgb"].mean()) + \
            (out["opacity"] ** 2 + 0.01).sqrt().mean() + ((out["sdf_grad"].norm(dim=-1) - 1) ** 2).mean()
        for p_ in [cache] + list(geo.parameters()):
            p_.grad = None
        loss.backward()
else:
    inp = bench.make_inputs(0, 1, dev, 1)
    rc = ops.RenderConfig()
    params = [inp["cache"]] + inp["sw"] + inp["fw"]

    def step():
        for t in params:
            t.grad = None
        out = functional.volume_render(inp["cache"], inp["sw"], inp["fw"], inp["ro"], inp["rd"], inp["ts"], inp["te"],
                                       inp["bg"], inp["cd"], inp["c2w"], rc, training=True)
        bench.loss_fn(out, inp["proj"]).backward()


lib = _lib.load()
lib.tt_tuning_phase_cycles.argtypes = [ctypes.c_void_p]
buf = (ctypes.c_uint64 * 40)()
lib.tt_tuning_phase_cycles(buf)  # allocate + reset
step()
lib.tt_tuning_phase_cycles(buf)  # warm-up discarded
n = 3
for _ in range(n):
    step()
lib.tt_tuning_phase_cycles(buf)
names = ["upstream cbar + skip test", "gather e (12 corners)", "park e in LDS", "k1 = relu(V1 e)", "k2 = relu(V2 k1)",
         "dV3 (transpose + VALU)", "k2bar, k1bar = V2^T k2bar", "dV1 outer products", "dV2 outer products",
         "ebar_p = V1_p^T k1bar + stage (x3)", "scatter_plane epilogue", "tail",
         "", "", "", "", "",  # (slots 12..16 carry the live-lane / scatter statistics printed above)
         "end of tile step -> pop", "item pop (queue atomic)", "ray set-up (issue)"]
geo_names = ["upstream (d sdf, d sdf_grad) + skip test", "gather f, u (12 corners)", "", "sdf net recompute + reverse chain (5 products)",
             "a1bar = W1 qbar, v", "a2bar = W2 b1bar, dw3 (transpose + VALU)", "", "dW1 outer products",
             "dW2 outer products", "scatter: q staging + corner set-up", "scatter epilogue", "tail",
             "", "", "", "", "", "  (of the epilogue) operands -> registers, next plane's set-up, operand split + MFMAs",
             "  (of the epilogue) next plane's slot claims", "  (of the epilogue) flush atomics, tag reset, lost references"]
print(f"texture backward: {buf[12] / n / 1024:.0f} live tile steps per wave per launch, {buf[13] / max(buf[12], 1):.1f} of 32 lanes live on average, {buf[14] / max(buf[12], 1):.1f} with |cbar| > 1e-12")
tex_pt, geo_pt = 3 * buf[12], buf[36]
print(f"scatter, texture kernel: {buf[15] / max(tex_pt, 1):.1f} active references per plane-tile, "
      f"{100.0 * buf[16] / max(buf[15], 1):.1f} % of them lost their slot (direct path)")
print(f"scatter, geometry kernel: {buf[34] / max(geo_pt, 1):.1f} active references per plane-tile, "
      f"{100.0 * buf[35] / max(buf[34], 1):.1f} % of them lost their slot (direct path)")
buf[12] = buf[13] = buf[14] = buf[15] = buf[16] = buf[34] = buf[35] = buf[36] = 0
for title, ofs, nms in (("k_decode_bwd_tex", 0, names), ("k_decode_bwd_geo", 20, geo_names)):
    tot = sum(buf[ofs:ofs + (17 if ofs == 20 else 20)])  # (geometry slots 17..19 are sub-phases OF the epilogue, not extra time)
    print(f"== {title}: {tot / n / 1024 / 1e6:.2f} M shader cycles per wave ==")
    for k, nm in enumerate(nms):
        if nm:
            print(f"{nm:48s} {buf[ofs + k] / n / 1024 / 1e3:9.1f} k cycles per wave   {100.0 * buf[ofs + k] / tot:5.1f} %")
