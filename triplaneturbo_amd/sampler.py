"""Sample placement along rays (no grad): uniform / stratified intervals and the one-level importance estimator
of the reference (threestudio/models/estimators.py:22-118 + prop_sigma_fn,
generative_space_sdf_volume_renderer.py:243-316), on the HIP kernels of csrc/tt_sampler.hip.

nerfacc's exact u-placement / jitter convention (pdf.cu of nerfacc v0.5.2) is not available, so WHERE the edges are
placed in cdf space is an explicit, switchable contract (`placement`, enum tt_sample_placement in include/tt_abi.h),
identical to oracle/cpu_ref.py::importance_sampling under either value:
 * "tt" (default): u_j = j / n, end points pinned to near / far; stratified: level-0 interior edges jittered by
   U(-.5,.5) cell, fine level u_j + U(0,1)/n clamped;
 * "center": u_j = (j + 0.5) / (n + 1) (centres of n + 1 equal cells); stratified: u_j = (j + U_j) / (n + 1);
and the rest is common to both:
 * level 0: n_prop intervals from the uniform cdf on [near, far];
 * proposal density: fixed-step NeuS density of the sdf at interval mid-points (sdf head only -- the reference
   also evaluates and discards the texture path here);
 * cdf = 1 - [T, 0], T = exp(-exclusive_cumsum(sigma * dt));
 * fine level: n_fine + 1 edges through the piecewise-linear inverse CDF; merged with the proposal edges in
   increasing order -> n_prop + n_fine + 1 intervals.
Random numbers are drawn here with torch (so a torch.Generator pins them) and handed to the kernels; everything
else -- edges, density, transmittance scan, inverse CDF, merge -- runs in two HIP kernels + the HIP decode."""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch

from . import ops

Tensor = torch.Tensor


def uniform_intervals(n_rays: int, n_samples: int, near: float, far: float, device=None, stratified: bool = False,
                      generator: Optional[torch.Generator] = None, placement: str = "tt") -> Tuple[Tensor, Tensor]:
    device = torch.device("cuda" if device is None else device)
    jitter = torch.rand(n_rays, n_samples + 1, device=device, generator=generator) if stratified else None
    return ops.sample_uniform(n_rays, n_samples, near, far, device, jitter, placement=placement)


@torch.no_grad()
def importance_sampling(sdf_fn: Callable[[Tensor, Tensor], Tensor], n_rays: int, n_prop: int, n_fine: int,
                        near: float, far: float, inv_std: float, render_step_size: float, device=None,
                        stratified: bool = False, generator: Optional[torch.Generator] = None,
                        placement: str = "tt", inv_std_t: Optional[Tensor] = None, use_volsdf: bool = False):
    """sdf_fn(t_starts, t_ends) -> sdf (n_rays, n_prop) at interval mid-points.  Returns t_starts, t_ends
    (n_rays, n_prop + n_fine + 1).  placement: "tt" | "center" (module docstring).  inv_std_t: device scalar that
    replaces the host float `inv_std` (trainable variance).  use_volsdf: VolSDF proposal density (renderer :286-287)."""
    device = torch.device("cuda" if device is None else device)
    ts, te = uniform_intervals(n_rays, n_prop, near, far, device, stratified, generator, placement)
    u = torch.rand(n_rays, n_fine + 1, device=device, generator=generator) if stratified else None
    return ops.sample_importance(ts, te, sdf_fn(ts, te), n_fine, inv_std, render_step_size, u, placement=placement,
                                 inv_std_t=inv_std_t, use_volsdf=use_volsdf)
