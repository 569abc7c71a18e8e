"""Renderer-level composite on top of the fused op: everything GenerativeSpaceSDFVolumeRenderer._forward does
with the per-ray accumulators (reference generative_space_sdf_volume_renderer.py:433-546).  These are a handful
of (n_rays, .)-sized torch ops; the per-sample hot path lives in the HIP kernels."""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import torch
import torch.nn.functional as F

from . import ops

Tensor = torch.Tensor


def _rot_world_to_cam(c2w: Tensor) -> Tensor:
    """inverse(c2w)[:, :3, :3] (renderer :478-479) for affine camera matrices (last row 0 0 0 1), as the closed-form
    inverse of the 3x3 block: rows of the inverse = cross products of the columns / det.  torch.inverse synchronises
    with the host (pivoting info) and cannot be captured in a hipGraph; this is a handful of element-wise kernels."""
    m = c2w[:, :3, :3]
    c0, c1, c2 = m[:, :, 0], m[:, :, 1], m[:, :, 2]
    r0, r1, r2 = torch.cross(c1, c2, dim=-1), torch.cross(c2, c0, dim=-1), torch.cross(c0, c1, dim=-1)
    det = (c0 * r0).sum(dim=-1, keepdim=True)
    return torch.stack([r0, r1, r2], dim=1) / det[:, :, None]


class LazyOutputs(dict):
    """The renderer's output dict.  The reference returns a dozen per-sample "training extras" on every call
    (renderer :532-545) although a given loss configuration reads only a few of them; here they are registered as
    thunks and materialised on first access (`out["normal"]`, `"normal" in out`, `.keys()`, `.items()` all behave
    like the eager dict; each thunk runs once, under the grad mode of the render call)."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self._lazy = {}

    def set_lazy(self, key, fn):
        self._lazy[key] = fn

    def _materialise(self, key):
        fn = self._lazy.pop(key)
        val = fn()
        super().__setitem__(key, val)
        return val

    def __getitem__(self, key):
        if key in self._lazy:
            return self._materialise(key)
        return super().__getitem__(key)

    def __setitem__(self, key, value):
        self._lazy.pop(key, None)
        super().__setitem__(key, value)

    def __contains__(self, key):
        return key in self._lazy or super().__contains__(key)

    def get(self, key, default=None):
        return self[key] if key in self else default

    def eager_keys(self):
        """keys whose values exist already (iteration / keys() would run every pending thunk)"""
        return list(super().keys())

    def _all(self):
        for k in list(self._lazy):
            self._materialise(k)

    def keys(self):
        self._all()
        return super().keys()

    def items(self):
        self._all()
        return super().items()

    def values(self):
        self._all()
        return super().values()

    def __iter__(self):
        self._all()
        return super().__iter__()

    def __len__(self):
        return super().__len__() + len(self._lazy)

    def update(self, *a, **k):
        for key, v in dict(*a, **k).items():
            self[key] = v


def volume_render(space_cache: Tensor, sdf_w: Sequence[Tensor], feat_w: Sequence[Tensor], rays_o: Tensor,
                  rays_d: Tensor, t_starts: Tensor, t_ends: Tensor, bg_color: Tensor, camera_distances: Tensor,
                  c2w: Tensor, rc: ops.RenderConfig, training: bool = True,
                  normal_direction: str = "camera", comp_rgb_bg: Optional[Tensor] = None,
                  packed: Optional[Tensor] = None, eval_termination_eps: float = 0.0) -> Dict[str, Tensor]:
    """rays_o/rays_d (B,H,W,3); t_starts/t_ends (B*H*W, S); bg_color (3,), (B*H*W,3) or (B,H,W,3);
    packed = ops.pack_planes(space_cache) when the caller already has it.
    Eval renders without autograd (training=False under no_grad) return no per-sample tensor, so they run on the fused
    decode + march kernel (tt_render_eval); eval_termination_eps > 0 lets it stop rays whose transmittance fell below
    it and skip texture decodes of weights below eps / S (per-ray error of opacity / rgb < 2 eps)."""
    B, Hh, Ww, _ = rays_o.shape
    n_rays = B * Hh * Ww
    S = t_starts.shape[1]
    ro = rays_o.reshape(n_rays, 3)
    rd = rays_d.reshape(n_rays, 3)
    if not training and not torch.is_grad_enabled():
        pk = packed if packed is not None else ops.planes_pack(space_cache)
        r = ops.render_eval_raw(pk, sdf_w, feat_w, ro.contiguous(), rd.contiguous(), t_starts.contiguous(),
                                t_ends.contiguous(), Hh * Ww, rc, image_w=Ww,
                                transmittance_eps=eval_termination_eps, weight_eps=eval_termination_eps / max(S, 1))
    else:
        r = ops.render_samples(space_cache, sdf_w, feat_w, ro, rd, t_starts, t_ends, Hh * Ww, rc, image_w=Ww,
                               packed=packed)
    opacity, depth, comp_rgb_fg, z_variance = r["opacity"], r["depth"], r["rgb_fg"], r["z_variance"]

    if bg_color.ndim == 1:
        bg = bg_color[None, :].expand(n_rays, -1)
    else:
        bg = bg_color.reshape(n_rays, -1)  # renderer :436-437
    comp_rgb = comp_rgb_fg + bg * (1.0 - opacity)  # :439
    if comp_rgb_bg is None:
        comp_rgb_bg = bg
    out = LazyOutputs({
        "comp_rgb": comp_rgb.view(B, Hh, Ww, -1),
        "comp_rgb_fg": comp_rgb_fg.view(B, Hh, Ww, -1),
        "comp_rgb_bg": comp_rgb_bg.reshape(B, Hh, Ww, -1),
        "opacity": opacity.view(B, Hh, Ww, 1),
        "depth": depth.view(B, Hh, Ww, 1),
        "z_variance": z_variance.view(B, Hh, Ww, 1),
    })
    # :452-462
    cd = camera_distances.reshape(-1, 1, 1, 1)
    far = cd + math.sqrt(3.0)
    near = cd - math.sqrt(3.0)
    disparity_tmp = out["depth"] * out["opacity"] + (1.0 - out["opacity"]) * far
    out["disparity"] = torch.clamp((far - disparity_tmp) / (far - near), 0.0, 1.0).view(B, Hh, Ww, 1)

    # :466-530
    comp_normal = F.normalize(r["normal_acc"], dim=-1)
    out["comp_normal"] = comp_normal.view(B, Hh, Ww, 3)
    if normal_direction == "camera":
        # (device-side constants only: no host scalars written into GPU tensors, so the step can be graph-captured)
        bg_normal = torch.cat([torch.full_like(comp_normal[:, :2], 0.5), torch.ones_like(comp_normal[:, :1])], dim=-1)
        bg_normal_white = torch.ones_like(comp_normal)
        rot = _rot_world_to_cam(c2w)
        comp_normal_cam = (comp_normal.view(B, -1, 3) @ rot.permute(0, 2, 1)).view(-1, 3)
        comp_normal_cam = torch.cat([-comp_normal_cam[:, :1], comp_normal_cam[:, 1:]], dim=-1)  # @ diag(-1, 1, 1)
        out["comp_normal_cam_vis"] = ((comp_normal_cam + 1.0) / 2.0 * opacity + (1 - opacity) * bg_normal).view(
            B, Hh, Ww, 3)
        out["comp_normal_cam_vis_white"] = (
            (comp_normal_cam + 1.0) / 2.0 * opacity + (1 - opacity) * bg_normal_white).view(B, Hh, Ww, 3)
    elif normal_direction == "front":
        n_prompts = space_cache.shape[0]
        nv = B // n_prompts
        bg_normal_white = torch.ones_like(comp_normal)
        c2w_front = c2w[0::nv].repeat_interleave(nv, dim=0)
        rot = _rot_world_to_cam(c2w_front)
        comp_normal_front = (comp_normal.view(B, -1, 3) @ rot.permute(0, 2, 1)).view(-1, 3)
        out["comp_normal_cam_vis_white"] = (
            (comp_normal_front + 1.0) / 2.0 * opacity + (1 - opacity) * bg_normal_white).view(B, Hh, Ww, 3)
    elif normal_direction != "world":
        raise ValueError(normal_direction)

    if training:  # :532-545 -- per-sample extras; kernel outputs are eager, derived tensors are lazy
        grad_mode = torch.is_grad_enabled()

        def lazy(fn):
            def run():
                with torch.set_grad_enabled(grad_mode):
                    return fn()
            return run

        cache = {}

        def shared(name, fn):
            if name not in cache:
                cache[name] = fn()
            return cache[name]

        t_pos = lambda: shared("t", lambda: ((t_starts + t_ends) / 2.0).reshape(-1, 1))
        ridx = lambda: shared("ri", lambda: torch.arange(n_rays, device=ro.device).unsqueeze(-1).expand(-1, S).reshape(-1))
        t_dirs = lambda: shared("td", lambda: rd[ridx()])
        points = lambda: shared("p", lambda: ro[ridx()] + t_dirs() * t_pos())
        normal = lambda: shared("n", lambda: F.normalize(r["sdf_grad"], dim=-1))
        out["weights"] = r["weights"]
        out["sdf"] = r["sdf"]
        out["features"] = r["features"]
        out["sdf_grad"] = r["sdf_grad"]
        out.set_lazy("t_points", lazy(t_pos))
        out.set_lazy("t_intervals", lazy(lambda: (t_ends - t_starts).reshape(-1, 1)))
        out.set_lazy("t_dirs", lazy(t_dirs))
        out.set_lazy("ray_indices", lazy(ridx))
        out.set_lazy("points", lazy(points))
        out.set_lazy("sdf_orig", lazy(lambda: r["sdf"] - ((points() ** 2).sum(dim=-1, keepdim=True).sqrt()
                                                          - rc.sdf_bias_radius)))
        out.set_lazy("normal", lazy(normal))
        out.set_lazy("shading_normal", lazy(normal))
        out["inv_std"] = torch.full((), float(rc.inv_std), device=ro.device)  # fill kernel, no host-to-device copy
    return out
