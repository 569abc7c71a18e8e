"""Synthetic inputs in the reference's training conventions (host-side helpers for bench.py / smoke()).

Camera -> ray conventions follow custom/triplaneturbo/data/multiview_multiprompt_multistep_datamodule_v2.py
:251-359 and threestudio/utils/ops.py:194-231,301-347: right-handed world (x back, y right, z up), camera looks at
the origin, up = +z, directions ((i+0.5-W/2)/f, -(j+0.5-H/2)/f, -1), f = 0.5*H/tan(fovy/2), rays_d normalised.
"""
from __future__ import annotations

import math
from typing import List, Sequence

import torch
import torch.nn.functional as F


def make_cameras(n_view: int, height: int, width: int, fovy_deg: float = 60.0, rel_distance: float = 0.9,
                 elevation_deg: float = 15.0, azimuth_start_deg: float = 0.0):
    """Returns rays_o, rays_d (n_view,H,W,3), c2w (n_view,4,4), camera_distances (n_view,) on the CPU."""
    fovy = torch.full((n_view,), math.radians(fovy_deg))
    azimuth = torch.deg2rad(azimuth_start_deg + 360.0 / n_view * torch.arange(n_view, dtype=torch.float32))
    elevation = torch.full((n_view,), math.radians(elevation_deg))
    dist = rel_distance / torch.tan(0.5 * fovy)  # relative_radius=True
    pos = torch.stack([dist * torch.cos(elevation) * torch.cos(azimuth),
                       dist * torch.cos(elevation) * torch.sin(azimuth),
                       dist * torch.sin(elevation)], dim=-1)
    world_up = torch.tensor([0.0, 0.0, 1.0]).expand(n_view, 3)
    lookat = F.normalize(-pos, dim=-1)
    right = F.normalize(torch.linalg.cross(lookat, world_up), dim=-1)
    up = F.normalize(torch.linalg.cross(right, lookat), dim=-1)
    c2w = torch.zeros(n_view, 4, 4)
    c2w[:, :3, 0], c2w[:, :3, 1], c2w[:, :3, 2], c2w[:, :3, 3] = right, up, -lookat, pos
    c2w[:, 3, 3] = 1.0
    focal = 0.5 * height / torch.tan(0.5 * fovy)
    px, py = torch.meshgrid(torch.arange(width, dtype=torch.float32) + 0.5,
                            torch.arange(height, dtype=torch.float32) + 0.5, indexing="xy")
    d = torch.stack([px - width / 2, -(py - height / 2), -torch.ones_like(px)], -1)[None].repeat(n_view, 1, 1, 1)
    d[..., :2] = d[..., :2] / focal[:, None, None, None]
    rays_d = F.normalize((d[:, :, :, None, :] * c2w[:, None, None, :3, :3]).sum(-1), dim=-1)
    rays_o = c2w[:, None, None, :3, 3].expand(rays_d.shape).contiguous()
    return rays_o, rays_d, c2w, dist


def init_mlp_weights(dims: Sequence[int], gen: torch.Generator) -> List[torch.Tensor]:
    """nn.Linear(bias=False) default init: U(-1/sqrt(fan_in), 1/sqrt(fan_in))."""
    out = []
    for fan_in, fan_out in zip(dims[:-1], dims[1:]):
        bound = 1.0 / math.sqrt(fan_in)
        out.append(((torch.rand(fan_out, fan_in, generator=gen, dtype=torch.float64) * 2 - 1) * bound).float())
    return out


def uniform_intervals(n_rays: int, n_samples: int, near: float, far: float):
    """n_samples equal intervals on [near, far] (level 0 of the reference's ImportanceEstimator with
    stratified=False, threestudio/models/estimators.py:61-79,104-118)."""
    s = torch.linspace(0.0, 1.0, n_samples + 1)
    t = (s * far + (1 - s) * near)[None, :].expand(n_rays, -1)
    return t[:, :-1].contiguous(), t[:, 1:].contiguous()
