"""The reference's training background, `multi-prompt-neural-hashgrid-environment-map-background`
(custom/triplaneturbo/models/background/multi_prompt_neural_environment_hashgrid_map_background.py:17-124): view
directions -> multiresolution hash encoding -> a two-layer MLP whose matrices a hyper-network produces from the
prompt embedding -> colour.  In the reference the encoding is tiny-cuda-nn's `HashGrid` (CUDA only); here it is the HIP
kernel pair tt_hashgrid_fwd / tt_hashgrid_bwd.  The hyper-network (`LinearHyperNetwork`,
geometry/hypernetwork.py:18-111) is three stock torch layers and stays torch, as does the per-prompt `bmm`.

Same registry name, Config fields, forward signature and state-dict keys (`encoding.encoding.encoding.params`,
`hypernet.layers.{0,1,3}.*`) as the reference.  Per-RAY work (n_rays points per render), not on the per-sample path."""
from __future__ import annotations

import ctypes
import random
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib
from .ops import _chk, _ptr, _stream
from .registry import BaseModule, register

Tensor = torch.Tensor


def _grid_cfg(config: dict) -> "_lib.HashGridCfg":
    if config.get("otype", "HashGrid") != "HashGrid":
        raise NotImplementedError("only the HashGrid encoding of the reference config is built")
    if config.get("interpolation", "Linear") != "Linear":
        raise NotImplementedError("HashGrid interpolation must be Linear (the tcnn default)")
    return _lib.HashGridCfg(int(config["n_levels"]), int(config["n_features_per_level"]),
                            int(config["log2_hashmap_size"]), int(config["base_resolution"]),
                            float(config["per_level_scale"]))


class _HashGridFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, params, cfg):
        x = _chk(x, "x")
        params = _chk(params, "params")
        n = x.shape[0]
        out = torch.empty((n, cfg.n_levels * cfg.n_features_per_level), device=x.device, dtype=torch.float32)
        st = _lib.load().tt_hashgrid_fwd(_ptr(x), n, _ptr(params), ctypes.byref(cfg), _ptr(out), _stream())
        _lib.check(st, "tt_hashgrid_fwd")
        ctx.save_for_backward(x)
        ctx.cfg, ctx.n_params = cfg, params.numel()
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_out):
        (x,) = ctx.saved_tensors
        grad = torch.zeros(ctx.n_params, device=x.device, dtype=torch.float32)
        st = _lib.load().tt_hashgrid_bwd(_ptr(x), x.shape[0], _ptr(g_out.contiguous()), ctypes.byref(ctx.cfg),
                                         _ptr(grad), _stream())
        _lib.check(st, "tt_hashgrid_bwd")
        return None, grad, None


class HashGrid(nn.Module):
    """tcnn.Encoding(3, {"otype": "HashGrid", ...}): `params` is the flat fp32 table in tcnn's layout, initialised
    U(-1e-4, 1e-4) like tcnn; input (N,3) in [0,1], output (N, n_levels * n_features_per_level) fp32."""

    def __init__(self, n_input_dims: int, config: dict):
        super().__init__()
        if n_input_dims != 3:
            raise NotImplementedError("3-D inputs only")
        self.n_input_dims = 3
        self.grid_cfg = _grid_cfg(config)
        n = _lib.load().tt_hashgrid_n_params(ctypes.byref(self.grid_cfg))
        if n < 0:
            _lib.check(int(n), "tt_hashgrid_n_params")
        self.params = nn.Parameter(torch.empty(int(n)).uniform_(-1e-4, 1e-4))
        self.n_output_dims = self.grid_cfg.n_levels * self.grid_cfg.n_features_per_level

    def forward(self, x: Tensor) -> Tensor:
        return _HashGridFn.apply(x.float(), self.params, self.grid_cfg)


class _Wrap(nn.Module):
    """One level of the reference's module nesting (CompositeEncoding -> TCNNEncoding -> tcnn.Encoding,
    threestudio/models/networks.py:17-64), kept so that state-dict keys match."""

    def __init__(self, inner: nn.Module):
        super().__init__()
        self.encoding = inner
        self.n_input_dims, self.n_output_dims = inner.n_input_dims, inner.n_output_dims

    def forward(self, x):
        return self.encoding(x)


def get_encoding(n_input_dims: int, config: dict) -> nn.Module:
    if config.get("include_xyz", False):
        raise NotImplementedError("include_xyz is not used by the reference background")
    return _Wrap(_Wrap(HashGrid(n_input_dims, config)))


class LinearHyperNetwork(nn.Module):
    """geometry/hypernetwork.py:18-111: c_dim -> [Linear(no bias), LayerNorm, SiLU] (+ hidden blocks with bias)
    -> Linear(bias) whose output is cut into the matrices of `out_dims` (each list prefixed by the encoding width)."""

    def __init__(self, n_input_dims: int, config: dict):
        super().__init__()
        if config.get("spectral_norm", False):
            raise NotImplementedError("spectral_norm=False in the reference config")
        if config.get("output_activation", None) not in (None, "none"):
            raise NotImplementedError("hypernet output_activation must be None (reference config)")
        self.out_dims: Dict[str, List[int]] = {}
        for key, val in dict(config.get("out_dims", {"bg_weights": [64, 3]})).items():
            val = list(val) if isinstance(val, (list, tuple)) or hasattr(val, "__iter__") else [val]
            self.out_dims[key] = [n_input_dims] + [int(v) for v in val]
        self.n_output_dims = sum(a * b for ch in self.out_dims.values() for a, b in zip(ch[:-1], ch[1:]))
        n_neurons, n_hidden = int(config["n_neurons"]), int(config["n_hidden_layers"])
        layers: List[nn.Module] = [self._linear(int(config["c_dim"]), n_neurons, bias=False), nn.LayerNorm(n_neurons),
                                   nn.SiLU(inplace=True)]
        for _ in range(n_hidden - 1):
            layers += [self._linear(n_neurons, n_neurons, bias=True), nn.LayerNorm(n_neurons), nn.SiLU(inplace=True)]
        layers += [self._linear(n_neurons, self.n_output_dims, bias=True)]
        self.layers = nn.Sequential(*layers)

    @staticmethod
    def _linear(dim_in: int, dim_out: int, bias: bool) -> nn.Linear:
        layer = nn.Linear(dim_in, dim_out, bias=bias)
        if bias:
            nn.init.zeros_(layer.bias)
        nn.init.xavier_normal_(layer.weight, gain=1.0)
        return layer

    def forward(self, x: Tensor) -> Dict[str, List[Tensor]]:
        out = self.layers(x.float())
        res: Dict[str, List[Tensor]] = {}
        start = 0
        for key, ch in self.out_dims.items():
            mats = []
            for a, b in zip(ch[:-1], ch[1:]):
                mats.append(out[:, start:start + a * b].reshape(*x.shape[:-1], a, b))
                start += a * b
            res[key] = mats
        return res


def _activation(name: Optional[str]):
    if name in (None, "none"):
        return lambda x: x
    if name == "sigmoid":
        return torch.sigmoid
    if name == "sigmoid-mipnerf":
        return lambda x: torch.sigmoid(x) * (1 + 2 * 0.001) - 0.001  # threestudio/utils/ops.py:118-119
    raise NotImplementedError(f"color_activation {name!r}")


@register("multi-prompt-neural-hashgrid-environment-map-background")
class MultipromptNeuralHashgridEnvironmentMapBackground(BaseModule):
    @dataclass
    class Config(BaseModule.Config):
        n_output_dims: int = 3
        color_activation: str = "sigmoid"
        pos_encoding_config: dict = field(default_factory=lambda: {
            "otype": "HashGrid", "n_levels": 8, "n_features_per_level": 2, "log2_hashmap_size": 19,
            "base_resolution": 4, "per_level_scale": 1.8114473285278132})  # desired resolution = 256
        hypernet_config: dict = field(default_factory=lambda: {
            "c_dim": 1024, "out_dims": {"bg_weights": [64, 3]}, "spectral_norm": False, "n_neurons": 64,
            "n_hidden_layers": 1, "output_activation": None})
        random_aug: bool = False
        random_aug_prob: float = 0.5
        random_aug_color: Optional[Tuple[float, float, float]] = None
        eval_color: Optional[Tuple[float, float, float]] = None

    cfg: Config

    def configure(self) -> None:
        self.encoding = get_encoding(3, dict(self.cfg.pos_encoding_config))
        self.hypernet = LinearHyperNetwork(self.encoding.n_output_dims, dict(self.cfg.hypernet_config))
        self.enabling_hypernet = True  # renderers pass text_embed when this is set (renderer :439-445)

    @staticmethod
    def hypernet_forward(enc: Tensor, params, activation=torch.relu, output_activation=None) -> Tensor:
        """:60-86 -- enc (B, HW, C); params: per-prompt matrices (P, C_in, C_out), B a multiple of P; no biases."""
        if torch.is_tensor(params):
            params = [params]
        for idx, p in enumerate(params):
            p = p.repeat_interleave(enc.shape[0] // p.shape[0], dim=0)
            enc = torch.bmm(enc, p)
            if activation is not None and idx < len(params) - 1:
                enc = activation(enc)
            elif output_activation is not None and idx == len(params) - 1:
                enc = output_activation(enc)
        return enc

    def forward(self, dirs: Tensor, text_embed: Optional[Tensor] = None) -> Tensor:
        """dirs (B,H,W,3) normalised view directions; text_embed (P, c_dim).  :88-124"""
        B, Hh, Ww, _ = dirs.shape
        if not self.training and self.cfg.eval_color is not None:
            return torch.ones(*dirs.shape[:-1], self.cfg.n_output_dims).to(dirs) * torch.as_tensor(
                self.cfg.eval_color).to(dirs)
        if text_embed is None:
            raise ValueError("this background is conditioned on the prompt: pass text_embed (P, c_dim)")
        bg_cache = self.hypernet(text_embed)
        d01 = (dirs + 1.0) / 2.0  # (-1, 1) => (0, 1)
        enc = self.encoding(d01.reshape(-1, 3))
        color = self.hypernet_forward(enc.view(B, Hh * Ww, -1), bg_cache["bg_weights"]).view(
            *dirs.shape[:-1], self.cfg.n_output_dims)
        color = _activation(self.cfg.color_activation)(color)
        if self.training and self.cfg.random_aug and random.random() < self.cfg.random_aug_prob:
            if self.cfg.random_aug_color is None:
                rc = torch.rand(B, 1, 1, self.cfg.n_output_dims)
            else:
                rc = torch.ones(B, 1, 1, self.cfg.n_output_dims) * torch.tensor(self.cfg.random_aug_color)
            color = color * 0 + rc.to(dirs).expand(*dirs.shape[:-1], -1)  # keeps every parameter in the graph
        return color
