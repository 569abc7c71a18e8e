"""Sample placement along rays (no grad): uniform / stratified intervals and the one-level importance
estimator of the reference (threestudio/models/estimators.py:22-118 + prop_sigma_fn,
generative_space_sdf_volume_renderer.py:243-316).

nerfacc's exact u-placement / jitter convention (pdf.cu of nerfacc v0.5.2) is not available, so this module
defines its own deterministic contract, identical to oracle/cpu_ref.py::importance_sampling:
 * level 0: n_prop equal intervals on [near, far] (stratified: every interior edge jittered by U(-.5,.5) cell);
 * proposal density: fixed-step NeuS density of the sdf at interval mid-points (HIP decode kernel, sdf head only --
   the reference also evaluates and discards the texture path here);
 * cdf = 1 - [T, 0], T = exp(-exclusive_cumsum(sigma * dt));
 * fine level: n_fine + 1 edges at u_k = k / n_fine (stratified: + U(0,1)/n_fine, clamped) through the
   piecewise-linear inverse CDF; merged with the proposal edges and sorted -> n_prop + n_fine + 1 intervals.
The torch ops below run on (n_rays, ~130)-sized tensors on the GPU; the per-point decode is the HIP kernel."""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch

Tensor = torch.Tensor


def uniform_intervals(n_rays: int, n_samples: int, near: float, far: float, device=None, stratified: bool = False,
                      generator: Optional[torch.Generator] = None) -> Tuple[Tensor, Tensor]:
    s = torch.linspace(0.0, 1.0, n_samples + 1, device=device)
    s = s[None, :].expand(n_rays, -1)
    if stratified:
        jitter = (torch.rand(n_rays, n_samples + 1, device=device, generator=generator) - 0.5) / n_samples
        jitter[:, 0] = 0
        jitter[:, -1] = 0
        s = s + jitter
    t = s * far + (1 - s) * near
    return t[:, :-1].contiguous(), t[:, 1:].contiguous()


def proposal_density(sdf: Tensor, inv_std: float, render_step_size: float) -> Tensor:
    """generative_space_sdf_volume_renderer.py:288-297"""
    prev_cdf = torch.sigmoid((sdf + render_step_size * 0.5) * inv_std)
    next_cdf = torch.sigmoid((sdf - render_step_size * 0.5) * inv_std)
    alpha = ((prev_cdf - next_cdf + 1e-5) / (prev_cdf + 1e-5)).clip(0.0, 1.0)
    return alpha / render_step_size


def importance_resample(t_edges: Tensor, cdfs: Tensor, n: int, stratified: bool = False,
                        generator: Optional[torch.Generator] = None) -> Tensor:
    R = t_edges.shape[0]
    u = torch.linspace(0.0, 1.0, n + 1, device=t_edges.device, dtype=t_edges.dtype)[None, :].expand(R, -1)
    if stratified:
        u = (u + torch.rand(R, n + 1, device=t_edges.device, generator=generator) / n).clamp(0.0, 1.0)
    u = u.contiguous()
    idx = torch.searchsorted(cdfs.contiguous(), u, right=True)
    lo = (idx - 1).clamp(0, cdfs.shape[1] - 1)
    hi = idx.clamp(0, cdfs.shape[1] - 1)
    c_lo, c_hi = cdfs.gather(1, lo), cdfs.gather(1, hi)
    t_lo, t_hi = t_edges.gather(1, lo), t_edges.gather(1, hi)
    denom = c_hi - c_lo
    frac = torch.where(denom > 0, (u - c_lo) / torch.where(denom > 0, denom, torch.ones_like(denom)),
                       torch.zeros_like(denom))
    return t_lo + frac.clamp(0, 1) * (t_hi - t_lo)


@torch.no_grad()
def importance_sampling(sdf_fn: Callable[[Tensor, Tensor], Tensor], n_rays: int, n_prop: int, n_fine: int,
                        near: float, far: float, inv_std: float, render_step_size: float, device=None,
                        stratified: bool = False, generator: Optional[torch.Generator] = None):
    """sdf_fn(t_starts, t_ends) -> sdf (n_rays, n_prop) at interval mid-points.  Returns t_starts, t_ends
    (n_rays, n_prop + n_fine + 1)."""
    ts, te = uniform_intervals(n_rays, n_prop, near, far, device, stratified, generator)
    t_vals = torch.cat([ts, te[:, -1:]], dim=1)
    sigma = proposal_density(sdf_fn(ts, te), inv_std, render_step_size)
    sd = sigma * (te - ts)
    excl = torch.cumsum(torch.cat([torch.zeros_like(sd[:, :1]), sd[:, :-1]], dim=1), dim=1)
    trans = torch.exp(-excl)
    cdfs = 1.0 - torch.cat([trans, torch.zeros_like(trans[:, :1])], dim=1)
    t_fine = importance_resample(t_vals, cdfs, n_fine, stratified, generator)
    t_all, _ = torch.sort(torch.cat([t_vals, t_fine], dim=1), dim=1)
    return t_all[:, :-1].contiguous(), t_all[:, 1:].contiguous()
