"""GPU test at the reference TRAINING shapes (configs/TriplaneTurbo_v1.yaml:8-9,133-150): 2 prompts x 4 views,
128x128 rays through PatchRenderer (42x42 global + 40x40 patch), importance sampling 128 + 64 -> 193 samples/ray,
planes 256^2.  Checks: runs, finite, sampler shape, gradients flow to planes and all six MLP matrices (the values
of the PatchRenderer composite are checked at a small size in test_gpu_plugin.py), and prints the timing."""
import time

import pytest
import torch

import triplaneturbo_amd as tt
from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu


def test_training_shapes_patch_renderer_importance_sampling():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    g = tt.find("few-step-triplane-dual-stable-diffusion")({}).to(dev)
    m = tt.find("no-material")({})
    b = tt.find("solid-color-background")({})
    base = dict(estimator="importance", trainable_variance=False, learned_variance_init=0.4605, num_samples_per_ray=64,
                num_samples_per_ray_importance=128, near_plane=0.1, far_plane=4.0,
                rgb_grad_shrink=[0, 1, 0.01, 20000], randomized=True)
    r = tt.find("patch-renderer")({"patch_size": 40, "global_downsample": 3,
                                   "base_renderer_type": "generative-space-sdf-volume-renderer",
                                   "base_renderer": base}, geometry=g, material=m, background=b).to(dev)
    r.train()
    r.update_step(0, 1000)
    gen = torch.Generator().manual_seed(1)
    P, n_view = 2, 4
    cache = (torch.randn(P, 6, 32, 256, 256, generator=gen) * 0.5).to(dev).requires_grad_(True)
    ro, rd, c2w, cd = O.make_cameras(P * n_view, 128, 128)
    kw = dict(space_cache=cache, text_embed=torch.zeros(P, 77, 1024), camera_distances=cd.to(dev), c2w=c2w.to(dev))
    ro, rd = ro.to(dev), rd.to(dev)

    def step():
        out = r(ro, rd, None, torch.ones(3, device=dev), **kw)
        loss = out["comp_rgb"].mean() + (out["opacity"] ** 2 + 0.01).sqrt().mean() + \
            ((out["sdf_grad"].norm(dim=-1) - 1) ** 2).mean() + out["comp_normal_cam_vis"].mean()
        for p_ in [cache] + list(g.parameters()):
            p_.grad = None
        loss.backward()
        return out, loss

    out, loss = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        out, loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    n_rays = P * n_view * (42 * 42 + 40 * 40)
    print(f"\nreference training shapes: {dt * 1e3:.1f} ms per PatchRenderer fwd+bwd "
          f"({n_rays} rays x 193 samples => {n_rays / dt / 1e6:.2f} M rays/s incl. importance sampling)")
    assert out["comp_rgb"].shape == (8, 128, 128, 3)
    # like the reference (`out = out_global`, patch_renderer.py:89) the per-sample extras are those of the GLOBAL
    # render: 8 views x 42x42 rays x 193 samples
    assert out["weights"].shape == (8 * 42 * 42 * 193, 1)
    assert torch.isfinite(loss) and torch.isfinite(cache.grad).all() and cache.grad.abs().sum() > 0
    for w in g.parameters():
        assert torch.isfinite(w.grad).all() and w.grad.abs().sum() > 0
