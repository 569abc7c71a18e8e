"""Decode half of the reference geometry plugin `few-step-triplane-dual-stable-diffusion`
(custom/triplaneturbo/models/geometry/few_step_triplane_dual_stable_diffusion.py:20-447): the two bias-free MLPs,
plane sampling and the per-point queries.  The SD-UNet/VAE triplane *generator* half stays stock PyTorch-ROCm and is
injected as `space_generator` (any object with forward / forward_denoise / forward_decode); it is out of scope.

Same registry name, Config fields, method names, argument meaning and state-dict keys
(`sdf_network.layers.{0,2,4}.weight`, `feature_network.layers.{0,2,4}.weight`) as the reference.
Per-point queries run on the HIP kernels (tt_query_points); the differentiable training path of the volume renderer
does not go through `forward` at all (it is fused in tt_render_fwd / tt_render_bwd_*)."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, Optional, Tuple, Union

import torch
import torch.nn as nn

from . import ops
from .registry import BaseModule, register

Tensor = torch.Tensor


class VanillaMLP(nn.Module):
    """threestudio/models/networks.py:67-104: Linear(bias=False) + ReLU, no output activation."""

    def __init__(self, dim_in: int, dim_out: int, config: dict):
        super().__init__()
        n_neurons, n_hidden = config["n_neurons"], config["n_hidden_layers"]
        if config.get("otype", "VanillaMLP") != "VanillaMLP" or config.get("output_activation", "none") not in (
                None, "none"):
            raise NotImplementedError("only otype=VanillaMLP, output_activation=none (the reference config)")
        layers = [nn.Linear(dim_in, n_neurons, bias=False), nn.ReLU(inplace=True)]
        for _ in range(n_hidden - 1):
            layers += [nn.Linear(n_neurons, n_neurons, bias=False), nn.ReLU(inplace=True)]
        layers += [nn.Linear(n_neurons, dim_out, bias=False)]
        self.layers = nn.Sequential(*layers)

    def weights(self):
        return [m.weight for m in self.layers if isinstance(m, nn.Linear)]


@register("few-step-triplane-dual-stable-diffusion")
class StableDiffusionTriplaneDualAttention(BaseModule):
    @dataclass
    class Config(BaseModule.Config):
        radius: float = 1.0
        n_feature_dims: int = 3
        space_generator_config: dict = field(default_factory=dict)
        mlp_network_config: dict = field(default_factory=lambda: {
            "otype": "VanillaMLP", "activation": "ReLU", "output_activation": "none", "n_neurons": 64,
            "n_hidden_layers": 2})
        backbone: str = "few_step_triplane_dual_stable_diffusion"
        normal_type: Optional[str] = "analytic"
        finite_difference_normal_eps: Union[float, str] = 0.01
        sdf_bias: Union[float, str] = "sphere"
        sdf_bias_params: Optional[Any] = 0.5
        isosurface_remove_outliers: bool = False
        isosurface_deformable_grid: bool = False
        rotate_planes: Optional[str] = "v1"
        split_channels: Optional[str] = "v1"
        geo_interpolate: str = "v1"
        tex_interpolate: str = "v2"

    cfg: Config

    def configure(self, space_generator: Optional[nn.Module] = None) -> None:
        c = self.cfg
        if c.rotate_planes != "v1" or c.geo_interpolate != "v1" or c.tex_interpolate != "v2" or \
                c.normal_type != "analytic" or c.sdf_bias != "sphere" or c.n_feature_dims != 3:
            raise NotImplementedError(
                "the HIP hot path implements the reference training config: rotate_planes=v1, geo_interpolate=v1, "
                "tex_interpolate=v2, normal_type=analytic, sdf_bias=sphere, n_feature_dims=3 "
                "(configs/TriplaneTurbo_v1.yaml:74-85)")
        if c.mlp_network_config["n_neurons"] != 64 or c.mlp_network_config["n_hidden_layers"] != 2:
            raise NotImplementedError("kernels are built for 64 neurons x 2 hidden layers")
        self.space_generator = space_generator
        self.sdf_network = VanillaMLP(32, 1, c.mlp_network_config)
        self.feature_network = VanillaMLP(96, c.n_feature_dims, c.mlp_network_config)
        if c.isosurface_deformable_grid:
            self.deformation_network = VanillaMLP(32, 3, c.mlp_network_config)
        self.register_buffer("bbox", torch.as_tensor([[-c.radius] * 3, [c.radius] * 3], dtype=torch.float32))
        # TT_Q_EXACT_F32 for every per-point query of this module (forward and backward kernels): all matrix products
        # on the fp32-input MFMA instead of split-fp16.  Not a reference knob, so not in Config.
        self.exact_f32 = False
        self.precision = None  # None / "split3" (default, fp32-grade) | "f32" | "split2" (fast): ops.RenderConfig.precision

    # ---- generator half: delegated (stock PyTorch-ROCm) ----
    def _gen(self):
        if self.space_generator is None:
            raise RuntimeError("no space_generator attached: the SD-UNet/VAE triplane generator is out of scope of "
                               "triplaneturbo_amd; pass one to the constructor or supply space_cache directly")
        return self.space_generator

    def generate_space_cache(self, styles, text_embed):
        return self._gen()(text_embed=text_embed, styles=styles)

    def denoise(self, noisy_input, text_embed, timestep):
        return self._gen().forward_denoise(text_embed=text_embed, noisy_input=noisy_input, t=timestep)

    def decode(self, latents):
        """few_step...:176-196: the generator's decoder emits 2C channels per plane; split_channels "v1" keeps the
        first C of the three geometry planes and the last C of the three texture planes."""
        triplane = self._gen().forward_decode(latents=latents)
        if self.cfg.split_channels is None:
            return triplane
        if self.cfg.split_channels != "v1":
            raise NotImplementedError(f"split_channels={self.cfg.split_channels!r}: only 'v1' / None")
        assert triplane.shape[1] == 6, "expected (B, 6, 2C, H, W)"
        c2 = triplane.shape[2] // 2
        return torch.cat([triplane[:, 0:3, :c2], triplane[:, 3:6, c2:]], dim=1)

    # ---- decode half ----
    def mlp_weights(self) -> Tuple[list, list]:
        return self.sdf_network.weights(), self.feature_network.weights()

    def _wants_grad(self, space_cache: Tensor, nets) -> bool:
        return torch.is_grad_enabled() and (space_cache.requires_grad or any(
            w.requires_grad for net in nets for w in net))

    def forward(self, points: Tensor, space_cache: Tensor, output_normal: bool = False) -> Dict[str, Tensor]:
        """few_step...:273-351.  points (B,N,3); space_cache (P,6,32,H,W) with B a multiple of P (view b reads
        prompt b // (B/P)).  Under autograd (training-time per-point decode, e.g. the raster path
        generative_space_mesh_rasterize_renderer.py:321-376) the outputs are connected to space_cache, the MLP weights
        and -- if they require grad, like positions interpolated from mesh vertices (:307-331) -- the points, second
        order through sdf_grad included."""
        B, N, _ = points.shape
        P = space_cache.shape[0]
        sw, fw = self.mlp_weights()
        pts_grad = torch.is_grad_enabled() and points.requires_grad
        pts = points.float() if pts_grad else points.detach().float()
        if pts_grad or self._wants_grad(space_cache, (sw, fw)):
            sdf, grad, feat = ops.query_points_grad(space_cache, sw, fw, pts, views_per_prompt=B // P,
                                                    radius=self.cfg.radius,
                                                    sdf_bias_radius=float(self.cfg.sdf_bias_params),
                                                    need_normal=output_normal, exact_f32=self.exact_f32, precision=self.precision)
        else:
            with torch.no_grad():
                packed = ops.planes_pack(space_cache.detach())
                sdf, grad, feat = ops.query_points(packed, [w.detach() for w in sw], [w.detach() for w in fw], pts,
                                                   views_per_prompt=B // P, radius=self.cfg.radius,
                                                   sdf_bias_radius=float(self.cfg.sdf_bias_params),
                                                   need_normal=output_normal, need_features=True,
                                                   exact_f32=self.exact_f32, precision=self.precision)
        bias = (pts.reshape(-1, 3) ** 2).sum(-1, keepdim=True).sqrt() - float(self.cfg.sdf_bias_params)
        out = {"sdf": sdf, "sdf_orig": sdf - bias, "features": feat}
        if output_normal:
            normal = torch.nn.functional.normalize(grad, dim=-1)
            out.update(normal=normal, shading_normal=normal, sdf_grad=grad)
        return out

    def forward_sdf(self, points: Tensor, space_cache: Tensor) -> Tensor:
        """few_step...:353-373.  Differentiable w.r.t. space_cache and the sdf net under autograd (like the
        reference); the points are constants."""
        B = points.shape[0]
        pts = points.reshape(B, -1, 3).detach().float()
        sw, fw = self.mlp_weights()
        kw = dict(views_per_prompt=B // space_cache.shape[0], radius=self.cfg.radius,
                  sdf_bias_radius=float(self.cfg.sdf_bias_params), exact_f32=self.exact_f32, precision=self.precision)
        if self._wants_grad(space_cache, (sw,)):
            # (the feature head is evaluated too and gets no upstream gradient: its backward is skipped)
            sdf, _, _ = ops.query_points_grad(space_cache, sw, fw, pts, need_normal=False, **kw)
        else:
            with torch.no_grad():
                sdf, _, _ = ops.query_points(ops.planes_pack(space_cache.detach()), [w.detach() for w in sw], None,
                                             pts, need_normal=False, need_features=False, **kw)
        return sdf.reshape(*points.shape[:-1], 1)

    def forward_field(self, points: Tensor, space_cache: Tensor):
        """few_step...:375-394: sdf (*N,1) and, with isosurface_deformable_grid, deformation (*N,3).  Under autograd
        (the mesh renderer's grid query in training, generative_space_mesh_rasterize_renderer.py:428-452) both are
        connected to space_cache, the sdf net and the deformation net."""
        if not self.cfg.isosurface_deformable_grid:
            return self.forward_sdf(points, space_cache), None
        B = points.shape[0]
        pts = points.reshape(B, -1, 3).detach().float()
        sw, _ = self.mlp_weights()
        dw = self.deformation_network.weights()
        kw = dict(views_per_prompt=B // space_cache.shape[0], radius=self.cfg.radius,
                  sdf_bias_radius=float(self.cfg.sdf_bias_params), exact_f32=self.exact_f32, precision=self.precision)
        if self._wants_grad(space_cache, (sw, dw)):
            sdf, deform = ops.query_field_grad(space_cache, sw, dw, pts, **kw)
        else:
            with torch.no_grad():
                sdf, deform = ops.query_field(ops.planes_pack(space_cache.detach()), [w.detach() for w in sw],
                                              [w.detach() for w in dw], pts, **kw)
        return sdf.reshape(*points.shape[:-1], 1), deform.reshape(*points.shape[:-1], 3)

    def forward_level(self, field: Tensor, threshold: float) -> Tensor:
        return field - threshold

    @torch.no_grad()
    def export(self, points: Tensor, space_cache: Tensor, **kwargs) -> Dict[str, Any]:
        """few_step...:402-430 (vertex colouring: feature net only)."""
        orig = points.shape
        out = self.forward(points.reshape(1, -1, 3), space_cache, output_normal=False)
        return {"features": out["features"].reshape(*orig[:-1], self.cfg.n_feature_dims)}
