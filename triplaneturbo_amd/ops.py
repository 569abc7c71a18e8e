"""Torch-facing wrappers over the C ABI (include/tt_abi.h).

PyTorch is plumbing here: it owns device memory and the current HIP stream; all arithmetic of the hot path
happens in libtt_hip.so.  Tensors must be CUDA(=HIP) fp32; there is no CPU fallback (a CPU tensor raises).
"""
from __future__ import annotations

import ctypes
import dataclasses
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib

Tensor = torch.Tensor


class KernelTimer:
    """Optional per-launch timing with HIP events on torch's current stream (the stream every kernel of this
    package is launched on).  bench.py uses it to report the dominant kernel's average duration."""

    def __init__(self):
        self.events = []  # (label, start, end)

    def summary(self, median: bool = False):
        """label -> (mean or median duration in ms, number of launches)"""
        torch.cuda.synchronize()
        acc = {}
        for label, a, b in self.events:
            acc.setdefault(label, []).append(a.elapsed_time(b))
        mid = (lambda v: sorted(v)[len(v) // 2]) if median else (lambda v: sum(v) / len(v))
        return {k: (mid(v), len(v)) for k, v in acc.items()}


_TIMER: Optional[KernelTimer] = None


def set_kernel_timer(t: Optional[KernelTimer]) -> None:
    global _TIMER
    _TIMER = t


class _timed:
    def __init__(self, label):
        self.label = label

    def __enter__(self):
        if _TIMER is not None:
            self.a = torch.cuda.Event(enable_timing=True)
            self.b = torch.cuda.Event(enable_timing=True)
            self.a.record()
        return self

    def __exit__(self, *exc):
        if _TIMER is not None:
            self.b.record()
            _TIMER.events.append((self.label, self.a, self.b))
        return False


def _ptr(t: Optional[Tensor]) -> ctypes.c_void_p:
    if t is None:
        return ctypes.c_void_p(0)
    return ctypes.c_void_p(t.data_ptr())


def _stream() -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t: Tensor, name: str, shape: Optional[Sequence[int]] = None, dtype=torch.float32) -> Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU (triplaneturbo_amd has no CPU path)")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if t.device.index != torch.cuda.current_device():
        # the kernels run on the current device's current stream (one process per GPU): a tensor of another GPU
        # would be a wild pointer there
        raise RuntimeError(f"{name} lives on {t.device} but the current device is cuda:{torch.cuda.current_device()}; "
                           f"call torch.cuda.set_device (one process per GPU)")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name} has shape {tuple(t.shape)}, expected {tuple(shape)}")
    return t.contiguous()


def _weights_struct(sdf_w: Sequence[Tensor], feat_w: Optional[Sequence[Tensor]]):
    sw = [_chk(sdf_w[0], "sdf w1", (64, 32)), _chk(sdf_w[1], "sdf w2", (64, 64)), _chk(sdf_w[2], "sdf w3", (1, 64))]
    fw: List[Optional[Tensor]] = [None, None, None]
    if feat_w is not None:
        fw = [_chk(feat_w[0], "feat v1", (64, 96)), _chk(feat_w[1], "feat v2", (64, 64)),
              _chk(feat_w[2], "feat v3", (3, 64))]
    st = _lib.MlpWeights(*[_ptr(t) for t in sw + fw])
    return st, sw + fw  # keep tensors alive


@dataclass
class RenderConfig:
    """Scalar knobs of GenerativeSpaceSDFVolumeRenderer that reach the kernels (reference renderer :40-71)."""
    radius: float = 1.0
    sdf_bias_radius: float = 0.5
    inv_std: float = 100.0
    cos_anneal_ratio: float = 1.0
    rgb_grad_shrink: float = 1.0
    tile_sb: int = 0  # consecutive samples of a ray per kernel tile (performance knob): 0 = default (2), 8 importance
    grad_copies: int = 1  # privatised copies of the plane-gradient buffer in the backward (performance knob)
    tile_chunk: int = 0  # samples of a ray block per work item (performance knob); 0 = automatic
    # precision of the MLP products (include/tt_abi.h): None / "split3" = fp32-grade three-piece products on the fp16 pipe
    # (default: the reference's precision, networks.py:91-97), "f32" = the fp32-input MFMA (A/B reference), "split2" /
    # "fast" = the two-piece fast mode of rounds 2-4 (~2^-21.5 per product).  exact_f32=True is the old spelling of "f32".
    precision: Optional[str] = None
    exact_f32: bool = False
    wgrad_f32: bool = False  # TT_R_WGRAD_F32: tuning build only (the product library rejects it)
    # TT_R_VOLSDF: alpha = |dists| x VolSDF density instead of the NeuS alpha (neus_volume_renderer.py:19-23,:95-96)
    use_volsdf: bool = False

    @property
    def prec(self) -> str:
        return _lib.resolve_precision(self.precision, self.exact_f32)
    # OPT-IN approximation of the backward (0 = exact): skip 32-sample tiles whose upstream gradients are all below the
    # threshold (tt_render_cfg.skip_eps_tex / skip_eps_geo in include/tt_abi.h; error measured in tests/test_gpu_skip.py)
    skip_eps_tex: float = 0.0
    skip_eps_geo: float = 0.0
    # trainable variance (reference class default, renderer :53,82): a 0-dim CUDA tensor holding inv_std (e.g.
    # exp(10 p).clamp(1e-6, 1e6)).  The kernels read it from the device (tt_render_cfg.inv_std_dev: no host read-back,
    # legal under stream capture) and, when it requires grad, render_samples returns d loss / d inv_std through autograd.
    # None = the host float `inv_std` above.
    inv_std_t: Optional[Tensor] = None
    # measurement hook (tt_render_cfg.stats): a zero-filled CUDA int64 tensor (3, 4); rows = forward / geometry backward /
    # texture backward decode kernel, columns = tile steps visited, tile steps executed, in-bounds (plane, sample) pairs,
    # reserved.  bench.py reads `live_tile_frac` from it; None = off
    stats: Optional[Tensor] = None


def planes_pack(space_cache: Tensor) -> Tensor:
    """(P,6,32,H,W) generator output -> (P,6,H,W,32) channels-last, rotate_planes 'v1' folded in."""
    space_cache = _chk(space_cache, "space_cache")
    if space_cache.ndim != 5 or space_cache.shape[1] != 6 or space_cache.shape[2] != 32:
        raise ValueError(f"space_cache must be (P,6,32,H,W), got {tuple(space_cache.shape)}")
    P, _, _, H, W = space_cache.shape
    out = torch.empty((P, 6, H, W, 32), device=space_cache.device, dtype=torch.float32)
    _lib.check(_lib.load().tt_planes_pack(_ptr(space_cache), _ptr(out), P, H, W, _stream()), "tt_planes_pack")
    return out


def planes_unpack_grad(grad_packed: Tensor) -> Tensor:
    """(P,6,H,W,32) or privatised (copies,P,6,H,W,32) packed gradients -> (P,6,32,H,W), copies summed."""
    grad_packed = _chk(grad_packed, "grad_packed")
    copies = grad_packed.shape[0] if grad_packed.ndim == 6 else 1
    P, _, H, W, _ = grad_packed.shape[-5:]
    out = torch.empty((P, 6, 32, H, W), device=grad_packed.device, dtype=torch.float32)
    _lib.check(_lib.load().tt_planes_unpack_grad(_ptr(grad_packed), _ptr(out), P, H, W, copies, _stream()),
               "tt_planes_unpack_grad")
    return out


class _PackPlanesFn(torch.autograd.Function):
    """tt_planes_pack as a differentiable op: its backward is tt_planes_unpack_grad (the exact transpose)."""

    @staticmethod
    def forward(ctx, space_cache):
        return planes_pack(space_cache)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_packed):
        return planes_unpack_grad(grad_packed.contiguous())


def pack_planes(space_cache: Tensor) -> Tensor:
    """Differentiable planes_pack.  A caller that renders the same cache several times (PatchRenderer: global + patch,
    plus the sampler's proposal pass) packs ONCE and hands `packed=` to render_samples / decode_rays: the packed
    gradients of all renders are summed by autograd and unpacked once (the reference re-materialises a rotated copy
    of the whole cache on every geometry call, few_step...:212-239)."""
    return _PackPlanesFn.apply(space_cache)


def query_points(packed: Tensor, sdf_w: Sequence[Tensor], feat_w: Optional[Sequence[Tensor]], points: Tensor,
                 views_per_prompt: int = 1, radius: float = 1.0, sdf_bias_radius: float = 0.5,
                 need_normal: bool = True, need_features: bool = True, exact_f32: bool = False,
                 precision: Optional[str] = None):
    """Per-point decode (no grad). points (B,N,3) -> sdf (B*N,1), sdf_grad (B*N,3)|None, features (B*N,3)|None."""
    packed = _chk(packed, "packed")
    points = _chk(points, "points")
    B, N, _ = points.shape
    P, _, H, W, _ = packed.shape
    wst, keep = _weights_struct(sdf_w, feat_w if need_features else None)
    dev = points.device
    sdf = torch.empty((B * N, 1), device=dev, dtype=torch.float32)
    grad = torch.empty((B * N, 3), device=dev, dtype=torch.float32) if need_normal else None
    feat = torch.empty((B * N, 3), device=dev, dtype=torch.float32) if need_features else None
    flags = (_lib.TT_Q_NORMAL if need_normal else 0) | (_lib.TT_Q_TEX if need_features else 0) | _lib.q_flag(
        _lib.resolve_precision(precision, exact_f32))
    with _timed("tt_query_points"):
        st = _lib.load().tt_query_points(_ptr(packed), ctypes.byref(wst), _ptr(points), B, N, P, views_per_prompt, H, W,
                                         radius, sdf_bias_radius, flags, _ptr(sdf), _ptr(grad), _ptr(feat), _stream())
    _lib.check(st, "tt_query_points")
    return sdf, grad, feat


def query_field(packed: Tensor, sdf_w: Sequence[Tensor], deform_w: Sequence[Tensor], points: Tensor,
                views_per_prompt: int = 1, radius: float = 1.0, sdf_bias_radius: float = 0.5, exact_f32: bool = False,
                precision: Optional[str] = None):
    """sdf (B*N,1) and deformation (B*N,3) from the geometry planes (forward_field)."""
    packed = _chk(packed, "packed")
    points = _chk(points, "points")
    B, N, _ = points.shape
    P, _, H, W, _ = packed.shape
    sw = [_chk(sdf_w[0], "sdf w1", (64, 32)), _chk(sdf_w[1], "sdf w2", (64, 64)), _chk(sdf_w[2], "sdf w3", (1, 64))]
    dw = [_chk(deform_w[0], "def d1", (64, 32)), _chk(deform_w[1], "def d2", (64, 64)),
          _chk(deform_w[2], "def d3", (3, 64))]
    wst = _lib.MlpWeights(*[_ptr(t) for t in sw + dw])
    sdf = torch.empty((B * N, 1), device=points.device, dtype=torch.float32)
    deform = torch.empty((B * N, 3), device=points.device, dtype=torch.float32)
    with _timed("tt_query_field"):
        st = _lib.load().tt_query_field(_ptr(packed), ctypes.byref(wst), _ptr(points), B, N, P, views_per_prompt, H, W,
                                        radius, sdf_bias_radius, _lib.q_flag(_lib.resolve_precision(precision, exact_f32)),
                                        _ptr(sdf), _ptr(deform), _stream())
    _lib.check(st, "tt_query_field")
    return sdf, deform


class _QueryPointsFn(torch.autograd.Function):
    """Differentiable per-point decode (geometry.forward in training, few_step...:273-351): forward =
    tt_planes_pack + tt_query_points, backward = tt_points_bwd_geo (sdf and, through the second-order chain,
    sdf_grad) + tt_points_bwd_tex (features) + tt_planes_unpack_grad, and tt_points_bwd_x for the query points.
    Differentiable inputs: space_cache, the six MLP matrices and the points (the reference keeps `points` in the
    graph: the raster renderer decodes positions interpolated from mesh vertices,
    generative_space_mesh_rasterize_renderer.py:307-331)."""

    @staticmethod
    def forward(ctx, space_cache, w1, w2, w3, v1, v2, v3, points, views_per_prompt, radius, sdf_bias_radius,
                need_normal, prec):
        ctx.set_materialize_grads(False)
        packed = planes_pack(space_cache)
        sdf, grad, feat = query_points(packed, (w1, w2, w3), (v1, v2, v3), points, views_per_prompt, radius,
                                       sdf_bias_radius, need_normal=need_normal, need_features=True,
                                       precision=prec)
        ctx.save_for_backward(packed, w1, w2, w3, v1, v2, v3, points)
        ctx.meta = (views_per_prompt, radius, sdf_bias_radius, _lib.q_flag(prec))
        if grad is None:
            grad = sdf.new_zeros((sdf.shape[0], 3))
            ctx.mark_non_differentiable(grad)
        return sdf, grad, feat

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_sdf, g_grad, g_feat):
        packed, w1, w2, w3, v1, v2, v3, points = ctx.saved_tensors
        vpp, radius, bias_r, qf = ctx.meta
        B, N, _ = points.shape
        P, _, H, W, _ = packed.shape
        wst, keep = _weights_struct((w1, w2, w3), (v1, v2, v3))
        lib = _lib.load()
        c = lambda t: None if t is None else t.contiguous()
        g_sdf, g_grad, g_feat = c(g_sdf), c(g_grad), c(g_feat)
        gw = [None] * 6
        g_cache = None
        if any(ctx.needs_input_grad[:7]):
            grad_packed = torch.zeros_like(packed)
            gw = [torch.zeros_like(t) for t in (w1, w2, w3, v1, v2, v3)]
            gst = _grads_struct(gw)
            if g_sdf is not None or g_grad is not None:
                ws = torch.empty((B * N, 4), device=packed.device, dtype=torch.float32)
                st = lib.tt_points_bwd_geo(_ptr(packed), ctypes.byref(wst), _ptr(points), B, N, P, vpp, H, W, radius,
                                           bias_r, qf, _ptr(g_sdf), _ptr(g_grad), _ptr(ws), _ptr(grad_packed),
                                           ctypes.byref(gst), _stream())
                _lib.check(st, "tt_points_bwd_geo")
            if g_feat is not None:
                st = lib.tt_points_bwd_tex(_ptr(packed), ctypes.byref(wst), _ptr(points), B, N, P, vpp, H, W, radius,
                                           3, qf, _ptr(g_feat), _ptr(grad_packed), ctypes.byref(gst), _stream())
                _lib.check(st, "tt_points_bwd_tex")
            g_cache = planes_unpack_grad(grad_packed) if ctx.needs_input_grad[0] else None
        g_points = None
        if ctx.needs_input_grad[7]:
            g_points = torch.empty_like(points)
            st = lib.tt_points_bwd_x(_ptr(packed), ctypes.byref(wst), _ptr(points), B, N, P, vpp, H, W, radius, qf,
                                     _ptr(g_sdf), _ptr(g_grad), _ptr(g_feat), _ptr(g_points), _stream())
            _lib.check(st, "tt_points_bwd_x")
        return (g_cache, *gw, g_points, None, None, None, None, None)


def query_points_grad(space_cache: Tensor, sdf_w: Sequence[Tensor], feat_w: Sequence[Tensor], points: Tensor,
                      views_per_prompt: int = 1, radius: float = 1.0, sdf_bias_radius: float = 0.5,
                      need_normal: bool = True, exact_f32: bool = False, precision: Optional[str] = None):
    """Differentiable per-point decode: sdf (B*N,1), sdf_grad (B*N,3) (zeros, non-differentiable, when
    need_normal is False), features (B*N,3); autograd-connected to space_cache, the six MLP matrices and -- when
    `points` requires grad -- the points.  precision / exact_f32: the mode of the forward and every backward kernel."""
    return _QueryPointsFn.apply(space_cache, sdf_w[0], sdf_w[1], sdf_w[2], feat_w[0], feat_w[1], feat_w[2],
                                _chk(points, "points"), int(views_per_prompt), float(radius), float(sdf_bias_radius),
                                bool(need_normal), _lib.resolve_precision(precision, bool(exact_f32)))


class _QueryFieldFn(torch.autograd.Function):
    """Differentiable implicit-field query (forward_field in training, few_step...:375-394; caller
    generative_space_mesh_rasterize_renderer.py:428-452): forward = tt_query_field, backward = tt_points_bwd_geo for
    the sdf head + tt_points_bwd_tex on the geometry planes for the deformation head (a 32->64->64->3 net on the SUM
    of the three planes is a 96->64->64->3 net with first-layer matrix [U1 U1 U1] on their concatenation)."""

    @staticmethod
    def forward(ctx, space_cache, w1, w2, w3, d1, d2, d3, points, views_per_prompt, radius, sdf_bias_radius,
                prec):
        ctx.set_materialize_grads(False)
        packed = planes_pack(space_cache)
        sdf, deform = query_field(packed, (w1, w2, w3), (d1, d2, d3), points, views_per_prompt, radius,
                                  sdf_bias_radius, precision=prec)
        ctx.save_for_backward(packed, w1, w2, w3, d1, d2, d3, points)
        ctx.meta = (views_per_prompt, radius, sdf_bias_radius, _lib.q_flag(prec))
        return sdf, deform

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_sdf, g_def):
        packed, w1, w2, w3, d1, d2, d3, points = ctx.saved_tensors
        vpp, radius, bias_r, qf = ctx.meta
        B, N, _ = points.shape
        P, _, H, W, _ = packed.shape
        d1x3 = d1.repeat(1, 3).contiguous()  # (64, 96) = [U1 U1 U1]
        keep = [w1.contiguous(), w2.contiguous(), w3.contiguous(), d1x3, d2.contiguous(), d3.contiguous()]
        wst = _lib.MlpWeights(*[_ptr(t) for t in keep])  # `keep` holds the contiguous copies until the launches
        grad_packed = torch.zeros_like(packed)
        gw = [torch.zeros_like(t) for t in (w1, w2, w3, d1x3, d2, d3)]
        gst = _grads_struct(gw)
        lib = _lib.load()
        if g_sdf is not None:
            ws = torch.empty((B * N, 4), device=packed.device, dtype=torch.float32)
            g_sdf = g_sdf.contiguous()
            st = lib.tt_points_bwd_geo(_ptr(packed), ctypes.byref(wst), _ptr(points), B, N, P, vpp, H, W, radius,
                                       bias_r, qf, _ptr(g_sdf), None, _ptr(ws), _ptr(grad_packed),
                                       ctypes.byref(gst), _stream())
            _lib.check(st, "tt_points_bwd_geo")
        if g_def is not None:
            g_def = g_def.contiguous()
            st = lib.tt_points_bwd_tex(_ptr(packed), ctypes.byref(wst), _ptr(points), B, N, P, vpp, H, W, radius, 0,
                                       qf, _ptr(g_def), _ptr(grad_packed), ctypes.byref(gst), _stream())
            _lib.check(st, "tt_points_bwd_tex")
        gw[3] = gw[3].view(64, 3, 32).sum(dim=1)
        g_cache = planes_unpack_grad(grad_packed) if ctx.needs_input_grad[0] else None
        return (g_cache, *gw, None, None, None, None, None)


def query_field_grad(space_cache: Tensor, sdf_w: Sequence[Tensor], deform_w: Sequence[Tensor], points: Tensor,
                     views_per_prompt: int = 1, radius: float = 1.0, sdf_bias_radius: float = 0.5,
                     exact_f32: bool = False, precision: Optional[str] = None):
    """Differentiable field query: sdf (B*N,1), deformation (B*N,3); autograd-connected to space_cache, the sdf net
    and the deformation net."""
    return _QueryFieldFn.apply(space_cache, sdf_w[0], sdf_w[1], sdf_w[2], deform_w[0], deform_w[1], deform_w[2],
                               _chk(points, "points"), int(views_per_prompt), float(radius), float(sdf_bias_radius),
                               _lib.resolve_precision(precision, bool(exact_f32)))


def _inv_std_args(rc: RenderConfig):
    """(host inv_std clamped like LearnedVariance.forward, renderer :34-35; device pointer of rc.inv_std_t or None).  ONE
    place validates the device scalar -- CUDA, fp32, current device, one element -- for every wrapper that hands it to a
    kernel (a CPU / fp64 / other-GPU tensor would be a wild pointer there)."""
    inv_std = min(max(float(rc.inv_std), 1.0e-6), 1.0e6)
    if rc.inv_std_t is None:
        return inv_std, None
    t = _chk(rc.inv_std_t, "inv_std_t")
    if t.numel() != 1:
        raise ValueError("inv_std_t must hold one float")
    if t.data_ptr() != rc.inv_std_t.data_ptr():
        raise ValueError("inv_std_t must be a contiguous float32 CUDA tensor (the kernels read it in place)")
    return inv_std, t.data_ptr()  # (the tensor is kept alive by `rc`, which the callers hold across the launch)


def _make_cfg(packed: Tensor, n_rays: int, rays_per_view: int, n_samples: int, rc: RenderConfig,
              per_sample: bool, image_w: int = 0, stats_row: int = 0) -> "_lib.RenderCfg":
    P, _, H, W, _ = packed.shape
    n_views = n_rays // rays_per_view
    if n_views * rays_per_view != n_rays or n_views % P != 0:
        raise ValueError(f"n_rays={n_rays} is not views*rays_per_view with views a multiple of P={P}")
    inv_std, inv_std_dev = _inv_std_args(rc)
    stats = None
    if rc.stats is not None:
        if rc.stats.dtype != torch.int64 or not rc.stats.is_cuda or tuple(rc.stats.shape) != (3, 4):
            raise ValueError("stats must be a CUDA int64 tensor of shape (3, 4)")
        stats = rc.stats.data_ptr() + 32 * stats_row
    return _lib.RenderCfg(P, n_views // P, H, W, rays_per_view, n_samples, n_rays, rc.radius, rc.sdf_bias_radius,
                          inv_std, rc.cos_anneal_ratio, rc.rgb_grad_shrink,
                          (_lib.TT_R_PER_SAMPLE if per_sample else 0) | _lib.r_flag(rc.prec) |
                          (_lib.TT_R_WGRAD_F32 if rc.wgrad_f32 else 0) |
                          (_lib.TT_R_VOLSDF if rc.use_volsdf else 0),
                          image_w if (image_w > 0 and rays_per_view % image_w == 0) else 0, int(rc.tile_sb),
                          max(1, int(rc.grad_copies)), max(0, int(rc.tile_chunk)), max(0.0, float(rc.skip_eps_tex)),
                          max(0.0, float(rc.skip_eps_geo)), inv_std_dev, stats)


def render_forward_raw(packed: Tensor, sdf_w: Sequence[Tensor], feat_w: Sequence[Tensor], rays_o: Tensor,
                       rays_d: Tensor, t_starts: Tensor, t_ends: Tensor, rays_per_view: int, rc: RenderConfig,
                       per_sample: bool = True, image_w: int = 0):
    """One tt_render_fwd call (decode kernel + march kernel).  rays_* (n_rays,3); t_* (n_rays,S); image_w = width
    of each view's ray image (enables 8x4 pixel tiles).  Returns a dict of raw kernel outputs."""
    packed = _chk(packed, "packed")
    rays_o, rays_d = _chk(rays_o, "rays_o"), _chk(rays_d, "rays_d")
    t_starts, t_ends = _chk(t_starts, "t_starts"), _chk(t_ends, "t_ends")
    n_rays, S = t_starts.shape
    if rays_o.shape != (n_rays, 3) or rays_d.shape != (n_rays, 3) or t_ends.shape != (n_rays, S):
        raise ValueError("ray / interval shapes disagree")
    cfg = _make_cfg(packed, n_rays, rays_per_view, S, rc, per_sample, image_w)
    wst, keep = _weights_struct(sdf_w, feat_w)
    dev = packed.device
    f32 = dict(device=dev, dtype=torch.float32)
    out = {
        "opacity": torch.empty((n_rays, 1), **f32), "depth": torch.empty((n_rays, 1), **f32),
        "rgb_fg": torch.empty((n_rays, 3), **f32), "z_variance": torch.empty((n_rays, 1), **f32),
        "normal_acc": torch.empty((n_rays, 3), **f32),
        "weights": torch.empty((n_rays * S, 1), **f32), "trans": torch.empty((n_rays * S, 1), **f32),
    }
    # per-sample decode results: outputs in training, inter-kernel workspace always
    out.update(sdf=torch.empty((n_rays * S, 1), **f32), sdf_grad=torch.empty((n_rays * S, 3), **f32),
               features=torch.empty((n_rays * S, 3), **f32))
    with _timed("tt_render_fwd"):
        st = _lib.load().tt_render_fwd(
            _ptr(packed), ctypes.byref(wst), _ptr(rays_o), _ptr(rays_d), _ptr(t_starts), _ptr(t_ends),
            ctypes.byref(cfg), _ptr(out["opacity"]), _ptr(out["depth"]), _ptr(out["rgb_fg"]),
            _ptr(out["z_variance"]), _ptr(out["normal_acc"]), _ptr(out["weights"]), _ptr(out["trans"]),
            _ptr(out.get("sdf")), _ptr(out.get("sdf_grad")), _ptr(out.get("features")), _stream())
    _lib.check(st, "tt_render_fwd")
    return out


@torch.no_grad()
def render_eval_raw(packed: Tensor, sdf_w: Sequence[Tensor], feat_w: Sequence[Tensor], rays_o: Tensor, rays_d: Tensor,
                    t_starts: Tensor, t_ends: Tensor, rays_per_view: int, rc: RenderConfig, image_w: int = 0,
                    transmittance_eps: float = 0.0, weight_eps: float = 0.0, stats: Optional[Tensor] = None):
    """tt_render_eval: the per-ray outputs of render_forward_raw (opacity, depth, rgb_fg, z_variance, normal_acc) from
    the fused decode + march kernel, no per-sample tensors, no autograd.  transmittance_eps / weight_eps > 0 switch
    on early termination / texture-decode skipping (error < transmittance_eps + S * weight_eps per ray); `stats` (2 x
    int64 on the device, zero-filled by the caller) receives the number of geometry / texture tile steps decoded."""
    packed = _chk(packed, "packed")
    rays_o, rays_d = _chk(rays_o, "rays_o"), _chk(rays_d, "rays_d")
    t_starts, t_ends = _chk(t_starts, "t_starts"), _chk(t_ends, "t_ends")
    n_rays, S = t_starts.shape
    if rays_o.shape != (n_rays, 3) or rays_d.shape != (n_rays, 3) or t_ends.shape != (n_rays, S):
        raise ValueError("ray / interval shapes disagree")
    cfg = _make_cfg(packed, n_rays, rays_per_view, S, rc, False, image_w)
    wst, keep = _weights_struct(sdf_w, feat_w)
    f32 = dict(device=packed.device, dtype=torch.float32)
    out = {"opacity": torch.empty((n_rays, 1), **f32), "depth": torch.empty((n_rays, 1), **f32),
           "rgb_fg": torch.empty((n_rays, 3), **f32), "z_variance": torch.empty((n_rays, 1), **f32),
           "normal_acc": torch.empty((n_rays, 3), **f32)}
    if stats is not None and (stats.dtype != torch.int64 or stats.numel() < 2 or not stats.is_cuda):
        raise ValueError("stats must be a CUDA int64 tensor with 2 elements")
    with _timed("tt_render_eval"):
        st = _lib.load().tt_render_eval(
            _ptr(packed), ctypes.byref(wst), _ptr(rays_o), _ptr(rays_d), _ptr(t_starts), _ptr(t_ends),
            ctypes.byref(cfg), float(transmittance_eps), float(weight_eps), _ptr(out["opacity"]), _ptr(out["depth"]),
            _ptr(out["rgb_fg"]), _ptr(out["z_variance"]), _ptr(out["normal_acc"]),
            ctypes.c_void_p(stats.data_ptr()) if stats is not None else ctypes.c_void_p(0), _stream())
    _lib.check(st, "tt_render_eval")
    return out


def _placement(name: str) -> int:
    if name not in _lib.PLACEMENTS:
        raise ValueError(f"placement must be one of {sorted(_lib.PLACEMENTS)}, got {name!r}")
    return _lib.PLACEMENTS[name]


@torch.no_grad()
def sample_uniform(n_rays: int, n_samples: int, near: float, far: float, device, jitter: Optional[Tensor] = None,
                   placement: str = "tt"):
    """tt_sample_uniform: level-0 intervals (n_rays, n_samples); jitter (n_rays, n_samples+1) U[0,1) => stratified;
    placement: "tt" | "center" (enum tt_sample_placement, include/tt_abi.h)."""
    device = torch.device(device)
    place = _placement(placement)
    if device.type != "cuda":
        raise RuntimeError("triplaneturbo_amd samplers run on the GPU only (no CPU fallback)")
    f32 = dict(device=device, dtype=torch.float32)
    ts, te = torch.empty((n_rays, n_samples), **f32), torch.empty((n_rays, n_samples), **f32)
    if jitter is not None:
        jitter = _chk(jitter, "jitter")
        if jitter.shape != (n_rays, n_samples + 1):
            raise ValueError("jitter must be (n_rays, n_samples + 1)")
    with torch.cuda.device(device):
        st = _lib.load().tt_sample_uniform(n_rays, n_samples, float(near), float(far), _ptr(jitter), place,
                                           _ptr(ts), _ptr(te), _stream())
    _lib.check(st, "tt_sample_uniform")
    return ts, te


@torch.no_grad()
def sample_importance(t_starts: Tensor, t_ends: Tensor, sdf: Tensor, n_fine: int, inv_std: float,
                      render_step_size: float, u_jitter: Optional[Tensor] = None, placement: str = "tt",
                      inv_std_t: Optional[Tensor] = None, use_volsdf: bool = False):
    """tt_sample_importance: proposal intervals (n_rays, K) + sdf at their mid-points -> (n_rays, K + n_fine + 1)
    intervals (proposal edges merged with n_fine + 1 inverse-CDF edges placed per `placement`).  inv_std_t: a 0-dim
    CUDA tensor that replaces the host float (trainable variance).  use_volsdf: the proposal density is the VolSDF
    density (renderer :286-287) instead of the fixed-step NeuS density (:288-297)."""
    place = _placement(placement) | (_lib.TT_PLACE_VOLSDF if use_volsdf else 0)
    if inv_std_t is not None:
        inv_std_t = _chk(inv_std_t.detach(), "inv_std_t")
    t_starts, t_ends, sdf = _chk(t_starts, "t_starts"), _chk(t_ends, "t_ends"), _chk(sdf, "sdf")
    n_rays, K = t_starts.shape
    if t_ends.shape != (n_rays, K) or sdf.numel() != n_rays * K:
        raise ValueError("proposal intervals / sdf shapes disagree")
    if u_jitter is not None:
        u_jitter = _chk(u_jitter, "u_jitter")
        if u_jitter.shape != (n_rays, n_fine + 1):
            raise ValueError("u_jitter must be (n_rays, n_fine + 1)")
    f32 = dict(device=t_starts.device, dtype=torch.float32)
    M = K + n_fine + 1
    ots, ote = torch.empty((n_rays, M), **f32), torch.empty((n_rays, M), **f32)
    with _timed("tt_sample_importance"):
        st = _lib.load().tt_sample_importance(_ptr(t_starts), _ptr(t_ends), _ptr(sdf), n_rays, K, int(n_fine),
                                              float(inv_std), _ptr(inv_std_t), float(render_step_size),
                                              _ptr(u_jitter), place,
                                              _ptr(ots), _ptr(ote), _stream())
    _lib.check(st, "tt_sample_importance")
    return ots, ote


@torch.no_grad()
def march_forward_raw(rays_d: Tensor, t_starts: Tensor, t_ends: Tensor, sdf: Tensor, sdf_grad: Tensor,
                      features: Tensor, rc: RenderConfig, out: Optional[dict] = None):
    """The ray march alone (tt_march_fwd): NeuS alpha, transmittance, weights and the five accumulations, on
    per-sample sdf (n_rays*S,1), sdf_grad (.,3), features (.,3) already decoded.  No grad (the differentiable path is
    render_samples).  `out` may hold preallocated result tensors (bench: re-time the march on live buffers)."""
    rays_d, t_starts, t_ends = _chk(rays_d, "rays_d"), _chk(t_starts, "t_starts"), _chk(t_ends, "t_ends")
    sdf, sdf_grad, features = _chk(sdf, "sdf"), _chk(sdf_grad, "sdf_grad"), _chk(features, "features")
    n_rays, S = t_starts.shape
    if sdf.numel() != n_rays * S or sdf_grad.numel() != 3 * n_rays * S or features.numel() != 3 * n_rays * S:
        raise ValueError("per-sample tensors do not match (n_rays, S)")
    inv_std, inv_std_dev = _inv_std_args(rc)
    cfg = _lib.RenderCfg(n_prompts=1, views_per_prompt=1, plane_h=1, plane_w=1, rays_per_view=n_rays, n_samples=S,
                         n_rays=n_rays, radius=rc.radius, sdf_bias_radius=rc.sdf_bias_radius, inv_std=inv_std,
                         cos_anneal_ratio=rc.cos_anneal_ratio, rgb_grad_shrink=rc.rgb_grad_shrink,
                         flags=_lib.TT_R_VOLSDF if rc.use_volsdf else 0, image_w=0,
                         tile_sb=0, grad_copies=1, tile_chunk=0, inv_std_dev=inv_std_dev)
    f32 = dict(device=rays_d.device, dtype=torch.float32)
    if out is None:
        out = {"opacity": torch.empty((n_rays, 1), **f32), "depth": torch.empty((n_rays, 1), **f32),
               "rgb_fg": torch.empty((n_rays, 3), **f32), "z_variance": torch.empty((n_rays, 1), **f32),
               "normal_acc": torch.empty((n_rays, 3), **f32), "weights": torch.empty((n_rays * S, 1), **f32),
               "trans": torch.empty((n_rays * S, 1), **f32)}
    with _timed("tt_march_fwd"):
        st = _lib.load().tt_march_fwd(_ptr(rays_d), _ptr(t_starts), _ptr(t_ends), ctypes.byref(cfg), _ptr(sdf),
                                      _ptr(sdf_grad), _ptr(features), _ptr(out["opacity"]), _ptr(out["depth"]),
                                      _ptr(out["rgb_fg"]), _ptr(out["z_variance"]), _ptr(out["normal_acc"]),
                                      _ptr(out["weights"]), _ptr(out["trans"]), _stream())
    _lib.check(st, "tt_march_fwd")
    return out


@torch.no_grad()
def march_backward_raw(rays_d: Tensor, t_starts: Tensor, t_ends: Tensor, fwd: dict, sdf: Tensor, sdf_grad: Tensor,
                       features: Tensor, rc: RenderConfig, g_opacity=None, g_depth=None, g_rgb_fg=None,
                       g_z_variance=None, g_normal_acc=None, g_weights=None, g_sdf=None, g_sdf_grad=None,
                       out: Optional[Tensor] = None, g_inv_std_rays: Optional[Tensor] = None):
    """tt_march_bwd: (n_rays*S, 4) = (d/d sdf, d/d sdf_grad) from upstream gradients of the march outputs; `fwd` is
    the dict march_forward_raw / render_forward_raw returned (opacity, depth, trans).  g_inv_std_rays (n_rays), if
    given, receives d loss / d inv_std per ray."""
    n_rays, S = t_starts.shape
    inv_std, inv_std_dev = _inv_std_args(rc)
    cfg = _lib.RenderCfg(n_prompts=1, views_per_prompt=1, plane_h=1, plane_w=1, rays_per_view=n_rays, n_samples=S,
                         n_rays=n_rays, radius=rc.radius, sdf_bias_radius=rc.sdf_bias_radius, inv_std=inv_std,
                         cos_anneal_ratio=rc.cos_anneal_ratio, rgb_grad_shrink=rc.rgb_grad_shrink,
                         flags=_lib.TT_R_VOLSDF if rc.use_volsdf else 0, image_w=0,
                         tile_sb=0, grad_copies=1, tile_chunk=0, inv_std_dev=inv_std_dev)
    if out is None:
        out = torch.empty((n_rays * S, 4), device=rays_d.device, dtype=torch.float32)
    c = lambda t: None if t is None else t.contiguous()
    gs = [c(t) for t in (g_opacity, g_depth, g_rgb_fg, g_z_variance, g_normal_acc, g_weights, g_sdf, g_sdf_grad)]
    with _timed("tt_march_bwd"):
        st = _lib.load().tt_march_bwd(_ptr(rays_d), _ptr(t_starts), _ptr(t_ends), ctypes.byref(cfg),
                                      _ptr(fwd["opacity"]), _ptr(fwd["depth"]), _ptr(fwd["trans"]), _ptr(sdf),
                                      _ptr(sdf_grad), _ptr(features), *[_ptr(t) for t in gs], _ptr(g_inv_std_rays),
                                      _ptr(out), _stream())
    _lib.check(st, "tt_march_bwd")
    return out


def _zeros_like_flat(tensors: Sequence[Tensor]) -> List[Tensor]:
    """zeros_like for a list of tensors as views of one buffer (one fill kernel instead of one per tensor)"""
    sizes = [t.numel() for t in tensors]
    flat = torch.zeros(sum(sizes), device=tensors[0].device, dtype=tensors[0].dtype)
    out, ofs = [], 0
    for t, n in zip(tensors, sizes):
        out.append(flat[ofs:ofs + n].view_as(t))
        ofs += n
    return out


def _grads_struct(tensors: Sequence[Tensor]):
    return _lib.MlpWeights(*[_ptr(t) for t in tensors])  # same layout as tt_mlp_grads (6 pointers)


class _TriplaneRenderFn(torch.autograd.Function):
    """Differentiable fused render on packed planes.  forward = tt_render_fwd; backward = tt_render_bwd_geo +
    tt_render_bwd_tex.  Differentiable inputs: the packed planes (see pack_planes) and the six MLP weights (sample
    positions are constants: the reference's sampler runs under no_grad, estimators.py:22)."""

    @staticmethod
    def forward(ctx, packed, w1, w2, w3, v1, v2, v3, rays_o, rays_d, t_starts, t_ends, rays_per_view, rc,
                image_w, inv_std_t=None):
        # inv_std_t: rc.inv_std_t again, as an autograd INPUT (trainable variance: its gradient is returned below)
        need_grad = any(ctx.needs_input_grad[:7]) or ctx.needs_input_grad[14]
        if inv_std_t is not None:  # keep a graph-free alias on ctx (same storage), not the autograd input itself
            rc = dataclasses.replace(rc, inv_std_t=inv_std_t.detach())
        ctx.set_materialize_grads(False)  # unused outputs (e.g. `features`, `weights`) reach backward as None, not as
        #                                    100 MB of zeros the kernels would have to read
        raw = render_forward_raw(packed, (w1, w2, w3), (v1, v2, v3), rays_o, rays_d, t_starts, t_ends, rays_per_view,
                                 rc, per_sample=True, image_w=image_w)
        ctx.rays_per_view = rays_per_view
        ctx.rc = rc
        ctx.image_w = image_w
        ctx.inv_std_shape = None if inv_std_t is None else tuple(inv_std_t.shape)
        if need_grad:
            ctx.save_for_backward(packed, w1, w2, w3, v1, v2, v3, rays_o, rays_d, t_starts, t_ends, raw["opacity"],
                                  raw["depth"], raw["trans"], raw["weights"], raw["features"], raw["sdf"],
                                  raw["sdf_grad"])
        ctx.mark_non_differentiable(raw["trans"])
        return (raw["opacity"], raw["depth"], raw["rgb_fg"], raw["z_variance"], raw["normal_acc"], raw["weights"],
                raw["sdf"], raw["sdf_grad"], raw["features"], raw["trans"])

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_op, g_depth, g_rgb, g_zvar, g_nacc, g_weights, g_sdf, g_sdf_grad, g_features, _g_trans):
        (packed, w1, w2, w3, v1, v2, v3, rays_o, rays_d, t_starts, t_ends, opacity, depth, trans, weights,
         features, sdf, sdf_grad) = ctx.saved_tensors
        n_rays, S = t_starts.shape
        cfg = _make_cfg(packed, n_rays, ctx.rays_per_view, S, ctx.rc, True, ctx.image_w, stats_row=1)
        cfg_tex = _make_cfg(packed, n_rays, ctx.rays_per_view, S, ctx.rc, True, ctx.image_w, stats_row=2)
        workspace = torch.empty((n_rays * S, 4), device=packed.device, dtype=torch.float32)
        wst, keep = _weights_struct((w1, w2, w3), (v1, v2, v3))
        copies = max(1, int(ctx.rc.grad_copies))
        grad_packed = torch.zeros((copies,) + tuple(packed.shape), device=packed.device, dtype=torch.float32)
        gw = _zeros_like_flat((w1, w2, w3, v1, v2, v3))  # six views of ONE zero-filled buffer: one fill launch
        gst = _grads_struct(gw)

        def c(t):
            return None if t is None else t.contiguous()

        g_op, g_depth, g_rgb, g_zvar, g_nacc = c(g_op), c(g_depth), c(g_rgb), c(g_zvar), c(g_nacc)
        g_weights, g_sdf, g_sdf_grad, g_features = c(g_weights), c(g_sdf), c(g_sdf_grad), c(g_features)
        lib = _lib.load()
        g_k_rays = None
        if ctx.needs_input_grad[14]:  # d loss / d inv_std, one partial per ray (summed below in a fixed order)
            g_k_rays = torch.empty((n_rays,), device=packed.device, dtype=torch.float32)
        with _timed("tt_render_bwd_geo"):
            st = lib.tt_render_bwd_geo(
                _ptr(packed), ctypes.byref(wst), _ptr(rays_o), _ptr(rays_d), _ptr(t_starts), _ptr(t_ends),
                ctypes.byref(cfg), _ptr(opacity), _ptr(depth), _ptr(trans), _ptr(sdf), _ptr(sdf_grad),
                _ptr(features), _ptr(g_op), _ptr(g_depth), _ptr(g_rgb), _ptr(g_zvar), _ptr(g_nacc), _ptr(g_weights),
                _ptr(g_sdf), _ptr(g_sdf_grad), _ptr(g_k_rays), _ptr(workspace), _ptr(grad_packed), ctypes.byref(gst),
                _stream())
        _lib.check(st, "tt_render_bwd_geo")
        with _timed("tt_render_bwd_tex"):
            st = lib.tt_render_bwd_tex(
                _ptr(packed), ctypes.byref(wst), _ptr(rays_o), _ptr(rays_d), _ptr(t_starts), _ptr(t_ends),
                ctypes.byref(cfg_tex), _ptr(weights), _ptr(features), _ptr(g_rgb), _ptr(g_features),
                _ptr(grad_packed), ctypes.byref(gst), _stream())
        _lib.check(st, "tt_render_bwd_tex")
        g_packed = None
        if ctx.needs_input_grad[0]:
            g_packed = grad_packed[0] if copies == 1 else grad_packed.sum(dim=0)
        g_inv_std = None
        if g_k_rays is not None:
            g_inv_std = g_k_rays.double().sum().float().reshape(ctx.inv_std_shape)
        return (g_packed, *gw, None, None, None, None, None, None, None, g_inv_std)


def render_samples(space_cache: Optional[Tensor], sdf_w: Sequence[Tensor], feat_w: Sequence[Tensor], rays_o: Tensor,
                   rays_d: Tensor, t_starts: Tensor, t_ends: Tensor, rays_per_view: int, rc: RenderConfig,
                   image_w: int = 0, packed: Optional[Tensor] = None):
    """Differentiable render for explicit sample intervals.  Returns a dict of per-ray accumulators and
    per-sample tensors (autograd-connected to space_cache and the MLP weights).  `packed` = pack_planes(space_cache)
    made by the caller (then space_cache is not looked at)."""
    names = ("opacity", "depth", "rgb_fg", "z_variance", "normal_acc", "weights", "sdf", "sdf_grad", "features",
             "trans")
    if packed is None:
        packed = pack_planes(space_cache)
    outs = _TriplaneRenderFn.apply(packed, sdf_w[0], sdf_w[1], sdf_w[2], feat_w[0], feat_w[1], feat_w[2],
                                   rays_o.contiguous(), rays_d.contiguous(), t_starts.contiguous(),
                                   t_ends.contiguous(), rays_per_view, rc, int(image_w), rc.inv_std_t)
    return dict(zip(names, outs))


@torch.no_grad()
def decode_rays(packed: Tensor, sdf_w: Sequence[Tensor], feat_w: Optional[Sequence[Tensor]], rays_o: Tensor,
                rays_d: Tensor, t_starts: Tensor, t_ends: Tensor, rays_per_view: int, rc: RenderConfig,
                need_normal: bool = False, need_features: bool = False, image_w: int = 0):
    """Per-sample decode along rays without the march (no grad): sdf (n_rays,S), optionally sdf_grad (n_rays,S,3)
    and features (n_rays,S,3).  The sampler's proposal pass uses the sdf-only form."""
    packed = _chk(packed, "packed")
    rays_o, rays_d = _chk(rays_o, "rays_o"), _chk(rays_d, "rays_d")
    t_starts, t_ends = _chk(t_starts, "t_starts"), _chk(t_ends, "t_ends")
    n_rays, S = t_starts.shape
    cfg = _make_cfg(packed, n_rays, rays_per_view, S, rc, False, image_w)
    wst, keep = _weights_struct(sdf_w, feat_w if need_features else None)
    f32 = dict(device=packed.device, dtype=torch.float32)
    sdf = torch.empty((n_rays, S), **f32)
    grad = torch.empty((n_rays, S, 3), **f32) if need_normal else None
    feat = torch.empty((n_rays, S, 3), **f32) if need_features else None
    flags = (_lib.TT_Q_NORMAL if need_normal else 0) | (_lib.TT_Q_TEX if need_features else 0)
    with _timed("tt_decode_rays"):
        st = _lib.load().tt_decode_rays(_ptr(packed), ctypes.byref(wst), _ptr(rays_o), _ptr(rays_d), _ptr(t_starts),
                                        _ptr(t_ends), ctypes.byref(cfg), flags, _ptr(sdf), _ptr(grad), _ptr(feat),
                                        _stream())
    _lib.check(st, "tt_decode_rays")
    return sdf, grad, feat


class _PatchCompositeFn(torch.autograd.Function):
    """tt_patch_composite_fwd / _bwd: bilinear upsample of the low-resolution global render + paste of the patch
    (patch_renderer.py:74-88) as one kernel each way."""

    @staticmethod
    def forward(ctx, low, patch, py, px, H, W, detach_low):
        low, patch = _chk(low, "low"), _chk(patch, "patch")
        B, h, w, C = low.shape
        PS = patch.shape[1]
        if patch.shape != (B, PS, PS, C):
            raise ValueError(f"patch {tuple(patch.shape)} does not match low {tuple(low.shape)}")
        out = torch.empty((B, H, W, C), device=low.device, dtype=torch.float32)
        st = _lib.load().tt_patch_composite_fwd(_ptr(low), _ptr(patch), _ptr(out), B, h, w, H, W, C, PS, py, px,
                                                _stream())
        _lib.check(st, "tt_patch_composite_fwd")
        ctx.meta = (B, h, w, H, W, C, PS, py, px, detach_low)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_out):
        B, h, w, H, W, C, PS, py, px, detach_low = ctx.meta
        g_out = g_out.contiguous()
        want_low = ctx.needs_input_grad[0] and not detach_low
        g_low = torch.empty((B, h, w, C), device=g_out.device, dtype=torch.float32) if want_low else None
        g_patch = torch.empty((B, PS, PS, C), device=g_out.device, dtype=torch.float32)
        st = _lib.load().tt_patch_composite_bwd(_ptr(g_out), _ptr(g_low), _ptr(g_patch), B, h, w, H, W, C, PS, py, px,
                                                _stream())
        _lib.check(st, "tt_patch_composite_bwd")
        return g_low, (g_patch if ctx.needs_input_grad[1] else None), None, None, None, None, None


def patch_composite(low: Tensor, patch: Tensor, py: int, px: int, H: int, W: int, detach_low: bool = False) -> Tensor:
    """(B,h,w,C) low-resolution image upsampled bilinearly (align_corners=False) to (B,H,W,C) with `patch`
    (B,PS,PS,C) pasted at rows py.., columns px..; differentiable w.r.t. both (low: unless detach_low)."""
    return _PatchCompositeFn.apply(low, patch, int(py), int(px), int(H), int(W), bool(detach_low))


_COMPOSITE_MODES = {"world": 0, "camera": 1, "front": 2}


class _CompositeFn(torch.autograd.Function):
    """tt_composite_fwd / _bwd: the renderer's per-ray composite (renderer :433-530) as one kernel each way.
    Differentiable inputs: opacity, depth, rgb_fg, normal_acc and the background colour (a learned background,
    background.py); the cameras are constants."""

    @staticmethod
    def forward(ctx, opacity, depth, rgb_fg, normal_acc, bg, cam_dist, c2w, rays_per_view, mode, view_group):
        opacity, depth = _chk(opacity, "opacity"), _chk(depth, "depth")
        rgb_fg, normal_acc = _chk(rgb_fg, "rgb_fg"), _chk(normal_acc, "normal_acc")
        bg, cam_dist = _chk(bg, "bg_color"), _chk(cam_dist, "camera_distances")
        n = opacity.numel()
        if rays_per_view <= 0 or n % rays_per_view != 0:
            raise ValueError(f"n_rays={n} is not a multiple of rays_per_view={rays_per_view}")
        views = n // rays_per_view
        if cam_dist.numel() != views:  # (composite() already expanded a 1-element tensor)
            raise ValueError(f"camera_distances has {cam_dist.numel()} elements, expected one per view ({views})")
        if c2w is None:
            if mode != 0:
                raise ValueError("c2w is required for normal_direction 'camera' / 'front' (renderer :478-530)")
        else:
            c2w = _chk(c2w, "c2w", (views, 4, 4))
        if view_group <= 0 or views % view_group != 0:
            raise ValueError(f"view_group={view_group} does not divide the number of views ({views})")
        bg_stride = 0 if bg.numel() == 3 else 3
        if bg_stride and bg.numel() != 3 * n:
            raise ValueError("bg_color must have 3 or 3 * n_rays elements")
        f32 = dict(device=opacity.device, dtype=torch.float32)
        comp_rgb, comp_normal = torch.empty((n, 3), **f32), torch.empty((n, 3), **f32)
        disparity = torch.empty((n, 1), **f32)
        vis = torch.empty((n, 3), **f32) if mode == 1 else None
        vis_white = torch.empty((n, 3), **f32) if mode != 0 else None
        st = _lib.load().tt_composite_fwd(_ptr(opacity), _ptr(depth), _ptr(rgb_fg), _ptr(normal_acc), _ptr(bg),
                                          bg_stride, _ptr(cam_dist), _ptr(c2w), n, rays_per_view, mode, view_group,
                                          _ptr(comp_rgb), _ptr(disparity), _ptr(comp_normal), _ptr(vis),
                                          _ptr(vis_white), _stream())
        _lib.check(st, "tt_composite_fwd")
        ctx.save_for_backward(opacity, depth, rgb_fg, normal_acc, bg, cam_dist, *(() if c2w is None else (c2w,)))
        ctx.meta = (n, bg_stride, rays_per_view, mode, view_group)
        ctx.set_materialize_grads(False)
        dummy = comp_rgb.new_zeros(0)
        outs = (comp_rgb, disparity, comp_normal, vis if vis is not None else dummy,
                vis_white if vis_white is not None else dummy)
        return outs

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_rgb, g_disp, g_cn, g_vis, g_visw):
        opacity, depth, rgb_fg, normal_acc, bg, cam_dist, *rest = ctx.saved_tensors
        c2w = rest[0] if rest else None
        n, bg_stride, rays_per_view, mode, view_group = ctx.meta
        c = lambda t: None if (t is None or t.numel() == 0) else t.contiguous()
        g_rgb, g_disp, g_cn, g_vis, g_visw = c(g_rgb), c(g_disp), c(g_cn), c(g_vis), c(g_visw)
        f32 = dict(device=opacity.device, dtype=torch.float32)
        g_op, g_dep = torch.empty((n, 1), **f32), torch.empty((n, 1), **f32)
        g_fg, g_na = torch.empty((n, 3), **f32), torch.empty((n, 3), **f32)
        g_bg = torch.empty((n, 3), **f32) if ctx.needs_input_grad[4] else None
        st = _lib.load().tt_composite_bwd(_ptr(opacity), _ptr(depth), _ptr(rgb_fg), _ptr(normal_acc), _ptr(bg),
                                          bg_stride, _ptr(cam_dist), _ptr(c2w), n, rays_per_view, mode, view_group,
                                          _ptr(g_rgb), _ptr(g_disp), _ptr(g_cn), _ptr(g_vis), _ptr(g_visw), _ptr(g_op),
                                          _ptr(g_dep), _ptr(g_fg), _ptr(g_na), _ptr(g_bg), _stream())
        _lib.check(st, "tt_composite_bwd")
        if g_bg is not None:
            g_bg = g_bg.sum(dim=0).view_as(bg) if bg_stride == 0 else g_bg.view_as(bg)
        return g_op, g_dep, g_fg, g_na, g_bg, None, None, None, None, None


def composite(opacity: Tensor, depth: Tensor, rgb_fg: Tensor, normal_acc: Tensor, bg_color: Tensor,
              camera_distances: Tensor, c2w: Optional[Tensor], rays_per_view: int, normal_direction: str = "camera",
              view_group: int = 1):
    """Per-ray composite of the renderer (renderer :433-530): returns comp_rgb (n,3), disparity (n,1), comp_normal
    (n,3), comp_normal_cam_vis (n,3) | None, comp_normal_cam_vis_white (n,3) | None.
    camera_distances: one per view, or a single element that is broadcast (as it broadcasts in the reference's
    `camera_distances.reshape(-1,1,1,1)` arithmetic); c2w (views,4,4), may be None for normal_direction 'world'."""
    mode = _COMPOSITE_MODES[normal_direction]
    views = opacity.numel() // max(int(rays_per_view), 1)
    camera_distances = camera_distances.reshape(-1).float()
    if camera_distances.numel() == 1 and views > 1:
        camera_distances = camera_distances.expand(views)
    camera_distances = camera_distances.contiguous()
    if c2w is not None:
        c2w = c2w.float().contiguous()
    rgb, disp, cn, vis, visw = _CompositeFn.apply(opacity, depth, rgb_fg, normal_acc, bg_color, camera_distances, c2w,
                                                  int(rays_per_view), mode, int(view_group))
    return rgb, disp, cn, (vis if mode == 1 else None), (visw if mode != 0 else None)


class _EikonalFn(torch.autograd.Function):
    """tt_eikonal_fwd / _bwd: mean((||sdf_grad|| - 1)^2) in one pass each way."""

    @staticmethod
    def forward(ctx, sdf_grad):
        g = _chk(sdf_grad, "sdf_grad")
        if g.ndim != 2 or g.shape[1] != 3:
            raise ValueError(f"sdf_grad must be (n, 3), got {tuple(g.shape)}")
        loss = torch.empty((), device=g.device, dtype=torch.float32)
        _lib.check(_lib.load().tt_eikonal_fwd(_ptr(g), g.shape[0], _ptr(loss), _stream()), "tt_eikonal_fwd")
        ctx.save_for_backward(g)
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_loss):
        g, = ctx.saved_tensors
        out = torch.empty_like(g)
        g_loss = g_loss.contiguous().float()
        _lib.check(_lib.load().tt_eikonal_bwd(_ptr(g), _ptr(g_loss), g.shape[0], _ptr(out), _stream()), "tt_eikonal_bwd")
        return out


def eikonal_loss(sdf_grad: Tensor) -> Tensor:
    """mean((||sdf_grad||_2 - 1)^2) over the samples (the training loop's eikonal regulariser,
    multiprompt_dual_renderer_multistep_generator.py:696-699) as one HIP kernel each way; sdf_grad (n,3) is the
    renderer's `out["sdf_grad"]`.  Equal to `((torch.linalg.norm(g, ord=2, dim=-1) - 1.0) ** 2).mean()`."""
    return _EikonalFn.apply(sdf_grad)
