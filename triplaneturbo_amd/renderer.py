"""`generative-space-sdf-volume-renderer` and `patch-renderer`: same registry names, Config fields, forward
signature, output-dict keys, update_step / train / eval behaviour as the reference
(custom/triplaneturbo/models/renderers/generative_space_sdf_volume_renderer.py:38-565,
threestudio/models/renderers/neus_volume_renderer.py:38-117, threestudio/models/renderers/patch_renderer.py).
All per-sample work runs in the HIP kernels; this file is orchestration."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, Optional

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional, ops, sampler
from .registry import BaseModule, C, find, register

Tensor = torch.Tensor


class LearnedVariance(nn.Module):
    """renderer :24-35"""

    def __init__(self, init_val, requires_grad=True):
        super().__init__()
        self.register_parameter("_inv_std", nn.Parameter(torch.tensor(float(init_val)), requires_grad=requires_grad))

    @property
    def inv_std(self):
        return torch.exp(self._inv_std * 10.0)

    def forward(self, x):
        return torch.ones_like(x) * self.inv_std.clamp(1.0e-6, 1.0e6)


@register("no-material")
class NoMaterial(BaseModule):
    """threestudio/models/materials/no_material.py: colour = activation(features).  The activation is fused into the
    HIP kernels, which implement the reference config's `sigmoid-mipnerf` only."""

    @dataclass
    class Config(BaseModule.Config):
        n_output_dims: int = 3
        color_activation: str = "sigmoid-mipnerf"
        input_feature_dims: Optional[int] = None
        mlp_network_config: Optional[dict] = None
        requires_normal: bool = False

    cfg: Config

    def configure(self) -> None:
        if self.cfg.color_activation != "sigmoid-mipnerf" or self.cfg.mlp_network_config is not None:
            raise NotImplementedError("fused material: color_activation=sigmoid-mipnerf without a material MLP "
                                      "(configs/TriplaneTurbo_v1.yaml:103-107)")
        self.requires_normal = self.cfg.requires_normal

    def forward(self, features: Tensor, **kwargs) -> Tensor:
        return torch.sigmoid(features) * (1 + 2 * 0.001) - 0.001


@register("solid-color-background")
class SolidColorBackground(BaseModule):
    """Constant colour per ray -- what the reference uses at eval time (`eval_color`, yaml :114) and what the
    benchmark uses.  The reference's training background is `background.py`."""

    @dataclass
    class Config(BaseModule.Config):
        color: tuple = (1.0, 1.0, 1.0)

    cfg: Config

    def forward(self, dirs: Tensor, **kwargs) -> Tensor:
        col = torch.as_tensor(self.cfg.color, device=dirs.device, dtype=dirs.dtype)
        return col.expand(*dirs.shape[:-1], 3)


@register("generative-space-sdf-volume-renderer")
class GenerativeSpaceSDFVolumeRenderer(BaseModule):
    @dataclass
    class Config(BaseModule.Config):
        radius: float = 1.0
        num_samples_per_ray: int = 512
        randomized: bool = True
        eval_chunk_size: int = 320000  # accepted, unused: the fused kernels never materialise per-layer activations
        learned_variance_init: float = 0.3
        cos_anneal_end_steps: int = 0
        use_volsdf: bool = False
        near_plane: float = 0.0
        far_plane: float = 1e10
        trainable_variance: bool = True
        estimator: str = "occgrid"
        grid_prune: bool = True
        prune_alpha_threshold: bool = True
        num_samples_per_ray_importance: int = 64
        train_chunk_size: int = 0  # accepted, unused (see eval_chunk_size)
        rgb_grad_shrink: Any = 1.0
        normal_direction: str = "camera"

    cfg: Config

    def configure(self, geometry, material, background) -> None:
        c = self.cfg
        if c.estimator != "importance":
            raise NotImplementedError("estimator must be 'importance' (the reference raises for 'occgrid', :83-85)")
        assert c.normal_direction in ["front", "camera", "world"]
        self.geometry, self.material, self.background = geometry, material, background
        if material is not None and not isinstance(material, NoMaterial):
            raise NotImplementedError("the fused path implements NoMaterial (sigmoid-mipnerf) only")
        # trainable_variance=True (the reference CLASS default, :53,82; the yaml switches it off, :136): the kernels read
        # inv_std from the device and tt_render_bwd_geo returns d loss / d inv_std (ops.RenderConfig.inv_std_t)
        self.variance = LearnedVariance(c.learned_variance_init, requires_grad=bool(c.trainable_variance))
        self.render_step_size = 1.732 * 2 * c.radius / c.num_samples_per_ray  # neus_volume_renderer.py:84-86
        self.cos_anneal_ratio = 1.0
        self.randomized = c.randomized
        self.rgb_grad_shrink = C(c.rgb_grad_shrink, 0, 0)
        self.register_buffer("bbox", torch.as_tensor([[-c.radius] * 3, [c.radius] * 3], dtype=torch.float32))
        # kernel tile shape under importance sampling (performance only; not a reference knob, so not in Config).
        # Measured at the reference training shapes (42^2 + 40^2 rays, 193 samples): sb 1/2/4/8/16 =
        # 14.4/12.9/12.5/12.3/12.8 ms per PatchRenderer forward+backward.
        self.tile_sb_importance = None  # None = by ray density (_tile_sb_for); an int pins it (performance only)
        # where the sampler places its edges in cdf space: "tt" | "center" (sampler.py; nerfacc's own convention is
        # unverifiable in this build, so the choice is explicit).  Not a reference knob, so not in Config.
        self.sampler_placement = "tt"
        # OPT-IN approximation of the training backward (0 = exact, like the reference): 32-sample tiles whose upstream
        # gradients are all below the threshold are skipped (ops.RenderConfig.skip_eps_tex / skip_eps_geo;
        # INTEGRATION.md section 5 for the measured error / speed).  Not reference knobs, so not in Config.
        self.grad_skip_eps_tex = 0.0
        self.grad_skip_eps_geo = 0.0
        # OPT-IN approximation for eval renders without autograd (tt_render_eval): eps > 0 stops marching a ray once its
        # transmittance is below eps and skips texture decodes of weights below eps / S; per-ray error of opacity and
        # comp_rgb < 2 eps, of depth < eps * far, of z_variance / comp_normal likewise O(eps) (measured at eps = 1e-4 on
        # the bench scene: opacity 1.0e-4, depth 1.7e-4, 17 % fewer geometry tile steps).
        # Default 0 = march every sample like the reference (renderer :317-324): a validation image rendered through
        # the plugin equals the reference's (it then runs on the training forward kernels, which are faster than the
        # fused kernel when nothing may be skipped: functional.volume_render).  Not a reference knob, so not in Config.
        self.eval_termination_eps = 0.0
        # precision of the MLP products (ops.RenderConfig.precision / include/tt_abi.h): None = "split3", the fp32-grade
        # default (the reference multiplies in fp32, networks.py:91-97); "f32" = fp32-input MFMA; "split2" = the fast mode.
        # Not a reference knob, so not in Config.
        self.precision = None

    # ------------------------------------------------------------------------------------------
    def _inv_std_value(self) -> float:
        """LearnedVariance.forward's value (exp(10 p) clamped to [1e-6, 1e6], renderer :29-35) as a host float.  The
        parameter is frozen (trainable_variance=False), so it is read back from the device only when it has been
        written (load_state_dict / manual assignment bump the tensor version): no host sync per render, and nothing
        that would be illegal under stream capture."""
        p = self.variance._inv_std
        key = (p._version, p.data_ptr())
        if getattr(self, "_inv_std_cache", (None, None))[0] != key:
            self._inv_std_cache = (key, min(max(math.exp(float(p.detach()) * 10.0), 1.0e-6), 1.0e6))
        return self._inv_std_cache[1]

    def set_inv_std(self, inv_std: float) -> None:
        """Write the variance parameter (inv_std = exp(10 p)) and refresh the cached host value.  Writes through
        `variance._inv_std.data` bypass the tensor version counter and are NOT seen by the cache: use this, assign
        with `copy_` / `load_state_dict` (both detected), or call `update_step` (which drops the cache)."""
        with torch.no_grad():
            self.variance._inv_std.fill_(math.log(float(inv_std)) / 10.0)
        self._inv_std_cache = (None, None)

    def _load_from_state_dict(self, *args, **kwargs):
        self._inv_std_cache = (None, None)
        return super()._load_from_state_dict(*args, **kwargs)

    def _render_config(self) -> ops.RenderConfig:
        g = self.geometry.cfg
        rc = ops.RenderConfig(radius=self.cfg.radius, sdf_bias_radius=float(g.sdf_bias_params),
                              cos_anneal_ratio=float(self.cos_anneal_ratio),
                              rgb_grad_shrink=float(self.rgb_grad_shrink), skip_eps_tex=float(self.grad_skip_eps_tex),
                              skip_eps_geo=float(self.grad_skip_eps_geo), precision=self.precision,
                              use_volsdf=bool(self.cfg.use_volsdf))
        if self.cfg.trainable_variance:
            # LearnedVariance.forward's value as a graph tensor on the device (renderer :29-35): the optimiser moves the
            # parameter every step, so nothing is read back to the host; rc.inv_std stays a placeholder the kernels ignore
            rc.inv_std_t = self.variance.inv_std.clamp(1.0e-6, 1.0e6).float()
        else:
            rc.inv_std = self._inv_std_value()
        return rc

    def sample(self, space_cache: Tensor, rays_o: Tensor, rays_d: Tensor, generator=None, packed=None):
        """ImportanceEstimator.sampling + prop_sigma_fn (estimators.py:22-101, renderer :243-316), no grad."""
        B, Hh, Ww, _ = rays_o.shape
        n_rays = B * Hh * Ww
        rc = self._render_config()
        sw, _ = self.geometry.mlp_weights()
        packed = ops.planes_pack(space_cache.detach()) if packed is None else packed.detach()
        ro, rd = rays_o.reshape(-1, 3).contiguous(), rays_d.reshape(-1, 3).contiguous()

        def sdf_fn(ts, te):
            sdf, _, _ = ops.decode_rays(packed, [w.detach() for w in sw], None, ro, rd, ts, te, Hh * Ww, rc,
                                        image_w=Ww)
            return sdf

        return sampler.importance_sampling(
            sdf_fn, n_rays, self.cfg.num_samples_per_ray_importance, self.cfg.num_samples_per_ray,
            self.cfg.near_plane, self.cfg.far_plane, rc.inv_std, self.render_step_size, device=rays_o.device,
            stratified=self.randomized, generator=generator, placement=self.sampler_placement,
            inv_std_t=rc.inv_std_t, use_volsdf=rc.use_volsdf)

    def forward(self, rays_o: Tensor, rays_d: Tensor, light_positions: Optional[Tensor] = None,
                bg_color: Optional[Tensor] = None, noise: Optional[Tensor] = None,
                space_cache: Optional[Tensor] = None, text_embed: Optional[Tensor] = None,
                camera_distances: Optional[Tensor] = None, c2w: Optional[Tensor] = None, t_starts=None, t_ends=None,
                **kwargs) -> Dict[str, Tensor]:
        """renderer :98-213.  rays (B,H,W,3); space_cache (P,6,32,R,R), B = P * n_view (view b uses prompt
        b // n_view -- the reference's repeat_interleave :120-143 is replaced by an index in the kernels, and its
        per-view eval loop :158-185 by the same single call).  Extra kwargs t_starts/t_ends (n_rays,S) bypass the
        sampler (used by the parity tests and the benchmark)."""
        B, Hh, Ww, _ = rays_o.shape
        if space_cache is None:
            space_cache = self.geometry.generate_space_cache(styles=noise, text_embed=text_embed)
        if not torch.is_tensor(space_cache):
            raise NotImplementedError("dict space caches (hyper-net variants, renderer :128-141) are out of scope")
        P = space_cache.shape[0]
        if text_embed is not None:
            assert text_embed.shape[0] == P
        assert B % P == 0, "batch of views must be a multiple of the number of prompts"
        grad_on = torch.is_grad_enabled()  # the reference stays differentiable in eval mode too
        packed = kwargs.pop("packed", None)  # PatchRenderer packs once for its two renders
        tile_sb = kwargs.pop("tile_sb", None)  # PatchRenderer: its sparse global render and its dense patch differ
        pitch_w = kwargs.pop("ray_pitch_w", None)  # width of the image whose pixel pitch these rays have (a crop: the full image)
        if packed is None:
            with (torch.enable_grad() if grad_on else torch.no_grad()):
                packed = ops.pack_planes(space_cache)
        importance_sampled = t_starts is None
        if t_starts is None:
            t_starts, t_ends = self.sample(space_cache, rays_o, rays_d, packed=packed)
        if bg_color is None:
            text_bg = kwargs.get("text_embed_bg", text_embed)
            comp_rgb_bg = self.background(dirs=rays_d, text_embed=text_bg) if getattr(
                self.background, "enabling_hypernet", False) else self.background(dirs=rays_d)
            bg_color = comp_rgb_bg
        else:
            comp_rgb_bg = None
        sw, fw = self.geometry.mlp_weights()
        rc = self._render_config()
        if importance_sampled:  # consecutive samples crowd into the same texels: 2x2-pixel x 8-sample tiles
            rc.tile_sb = int(tile_sb) if tile_sb is not None else self._tile_sb_for(
                int(pitch_w) if pitch_w else rays_o.shape[2], space_cache.shape[-1])
        ctx = torch.enable_grad() if grad_on else torch.no_grad()
        with ctx:
            out = functional.volume_render(space_cache, sw, fw, rays_o, rays_d, t_starts, t_ends, bg_color,
                                           camera_distances, c2w, rc, training=self.training,
                                           normal_direction=self.cfg.normal_direction, comp_rgb_bg=comp_rgb_bg,
                                           packed=packed, eval_termination_eps=self.eval_termination_eps)
        if self.training:
            out["inv_std"] = self.variance.inv_std
        return out

    def _tile_sb_for(self, image_w: int, plane_w: int) -> int:
        """Kernel tile shape under importance sampling (results do not depend on it).  Consecutive fine samples of a ray
        crowd into the same texels, so sparse ray grids (adjacent pixels several texels apart: PatchRenderer's two
        renders) want 2x2-pixel x 8-sample tiles; once adjacent pixels are less than about a texel apart (image at least
        as wide as the planes) the 4x4-pixel x 2-sample tile of uniform sampling shares more texels: measured on
        configs[1] with the 128 + 64 sampler, 256^2 rays on 256^2 planes, 12.8 (sb 2) / 13.8 (4) / 14.9 (8) / 17.2 (16)
        ms per step; training shapes (42^2 + 40^2 rays) 9.0 / 8.8 / 8.8 / 8.9 (tools/time_training_shapes.py)."""
        if self.tile_sb_importance is not None:
            return int(self.tile_sb_importance)
        # pixel pitch in texels.  Round 6 (dense-rank scatter, 128 distinct texels per plane-tile): swept again at the
        # training shapes, global render (pitch 6) x patch (a crop of the full-resolution image: pitch 2) -- 8 / 4 is the
        # best pair (8.48 ms against 8.66 for 8 / 8 and 8.62 for 4 / 4; profiles/r06_tile_sb_sweep.txt)
        pitch = plane_w / max(image_w, 1)
        return 2 if pitch <= 1.0 else (4 if pitch <= 3.0 else 8)

    def update_step(self, epoch: int, global_step: int, on_load_weights: bool = False) -> None:
        self._inv_std_cache = (None, None)  # one read-back per step at most; also catches `.data` writes
        self.rgb_grad_shrink = C(self.cfg.rgb_grad_shrink, epoch, global_step)  # renderer :548-553
        self.cos_anneal_ratio = 1.0 if self.cfg.cos_anneal_end_steps == 0 else min(
            1.0, global_step / self.cfg.cos_anneal_end_steps)  # neus_volume_renderer.py:385-389

    def train(self, mode=True):
        self.randomized = mode and self.cfg.randomized
        if hasattr(self.geometry, "train"):
            self.geometry.train(mode)
        return super().train(mode=mode)

    def eval(self):
        self.randomized = False
        if hasattr(self.geometry, "eval"):
            self.geometry.eval()
        return super().eval()


@register("patch-renderer")
class PatchRenderer(BaseModule):
    """threestudio/models/renderers/patch_renderer.py: training = low-res global render + random hi-res patch."""

    @dataclass
    class Config(BaseModule.Config):
        radius: float = 1.0
        patch_size: int = 128
        base_renderer_type: str = ""
        base_renderer: Optional[dict] = None
        global_detach: bool = False
        global_downsample: int = 4

    cfg: Config

    def configure(self, geometry, material, background) -> None:
        self.base_renderer = find(self.cfg.base_renderer_type)(self.cfg.base_renderer, geometry=geometry,
                                                               material=material, background=background)
        # kernel tile shapes of the two renders (performance only; None = the base renderer's choice)
        self.tile_sb_global = None
        self.tile_sb_patch = None

    def forward(self, rays_o: Tensor, rays_d: Tensor, light_positions: Optional[Tensor] = None,
                bg_color: Optional[Tensor] = None, **kwargs) -> Dict[str, Tensor]:
        B, H, W, _ = rays_o.shape
        if not self.base_renderer.training:
            return self.base_renderer(rays_o, rays_d, light_positions, bg_color, **kwargs)
        sc = kwargs.get("space_cache")
        if torch.is_tensor(sc) and kwargs.get("packed") is None:  # one pack (and one gradient unpack) for both renders
            with (torch.enable_grad() if torch.is_grad_enabled() else torch.no_grad()):
                kwargs["packed"] = ops.pack_planes(sc)
        ds = self.cfg.global_downsample
        # (channels-first COPIES: ATen's bilinear kernel is ~10x slower on the permuted view -- 142 us per call at 32 views of
        # 128 x 128, 1 % of a training-shape pass -- and computes the same values)
        g_o = F.interpolate(rays_o.permute(0, 3, 1, 2).contiguous(), (H // ds, W // ds), mode="bilinear").permute(0, 2, 3, 1)
        g_d = F.interpolate(rays_d.permute(0, 3, 1, 2).contiguous(), (H // ds, W // ds), mode="bilinear").permute(0, 2, 3, 1)
        # (performance hints only: the patch is a crop of the full-resolution image, its rays keep that image's pixel pitch)
        kw_g = dict(kwargs, ray_pitch_w=W // ds) if self.tile_sb_global is None else dict(kwargs, tile_sb=self.tile_sb_global)
        kw_p = dict(kwargs, ray_pitch_w=W) if self.tile_sb_patch is None else dict(kwargs, tile_sb=self.tile_sb_patch)
        out_global = self.base_renderer(g_o.contiguous(), g_d.contiguous(), light_positions, bg_color, **kw_g)
        PS = self.cfg.patch_size
        px = torch.randint(0, W - PS, (1,)).item()
        py = torch.randint(0, H - PS, (1,)).item()
        out = self.base_renderer(rays_o[:, py:py + PS, px:px + PS].contiguous(),
                                 rays_d[:, py:py + PS, px:px + PS].contiguous(), light_positions, bg_color, **kw_p)
        eager = out.eager_keys() if hasattr(out, "eager_keys") else list(out)  # per-sample extras stay lazy
        valid = [k for k in eager if torch.is_tensor(out[k]) and out[k].ndim == out["comp_rgb"].ndim
                 and out[k][..., 0].shape == out["comp_rgb"][..., 0].shape]
        grad_mode = torch.is_grad_enabled()
        detach = self.cfg.global_detach

        def composite(low: Tensor, patch: Tensor):
            # bilinear upsample of the global render + paste of the patch (patch_renderer.py:74-88) as ONE HIP kernel
            # (tt_patch_composite_fwd; its backward is one more), evaluated when the key is first read: a loss reads a
            # few of the ~10 image-shaped outputs, the reference upsamples all of them with six torch ops each
            def run():
                with torch.set_grad_enabled(grad_mode):
                    return ops.patch_composite(low, patch, py, px, H, W, detach_low=detach)
            return run

        if not isinstance(out_global, functional.LazyOutputs):
            out_global = functional.LazyOutputs(out_global)
        for k in valid:
            out_global.set_lazy(k, composite(dict.__getitem__(out_global, k), out[k]))
        return out_global

    def update_step(self, epoch: int, global_step: int, on_load_weights: bool = False) -> None:
        self.base_renderer.update_step(epoch, global_step, on_load_weights)

    def train(self, mode=True):
        return self.base_renderer.train(mode)

    def eval(self):
        return self.base_renderer.eval()
