"""Generate golden vectors under tests/golden/ (run ONCE in the build container).

    python tests/golden/make_golden.py

Part A (G1-G4) imports the dependency-light copies of the reference's hot-path
functions straight from /root/reference (read-only, never copied):
  triplaneturbo_executable/utils/general_utils.py  (sample_from_planes, project_onto_planes, scale_tensor ...)
  triplaneturbo_executable/models/networks.py      (VanillaMLP, get_activation)
which are verbatim copies of custom/triplaneturbo/models/geometry/utils.py:31-145
and threestudio/models/networks.py:67-104.  Two 3-line annotation stubs
(jaxtyping.Float, omegaconf.OmegaConf) stand in for import-time-only packages
that are not installed; they carry no arithmetic.

Part B (G5-G7) dumps the oracle's own full render / backward / grad2 vectors in
fp64 and fp32 so the GPU box (which has no /root/reference) can replay them.

Only DATA is written (inputs + expected outputs); no reference source travels.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle import cpu_ref as O  # noqa: E402


def _load_reference():
    # annotation-only stubs
    jt = types.ModuleType("jaxtyping")

    class _Ann:
        def __getitem__(self, item):
            return torch.Tensor

    jt.Float = _Ann()
    sys.modules.setdefault("jaxtyping", jt)
    oc = types.ModuleType("omegaconf")

    class OmegaConf:  # only referenced inside config_to_primitive (unused here)
        @staticmethod
        def to_container(c, resolve=True):
            return c

    oc.OmegaConf = OmegaConf
    sys.modules.setdefault("omegaconf", oc)

    def load(name, path, package=None):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        return mod

    # fake package skeleton so the relative import in networks.py resolves without
    # executing triplaneturbo_executable/__init__.py (which pulls diffusers)
    pkg = types.ModuleType("tte"); pkg.__path__ = []
    sys.modules["tte"] = pkg
    for sub in ("utils", "models"):
        m = types.ModuleType(f"tte.{sub}"); m.__path__ = []
        sys.modules[f"tte.{sub}"] = m
    gu = load("tte.utils.general_utils", f"{REF}/triplaneturbo_executable/utils/general_utils.py")
    nw = load("tte.models.networks", f"{REF}/triplaneturbo_executable/models/networks.py")
    return gu, nw


def main():
    gu, nw = _load_reference()
    g = torch.Generator().manual_seed(1234)
    out = {}

    # ---------------- G1: sample_from_planes v1 + v2 (imported reference) --------------
    planes = torch.randn(2, 3, 8, 16, 16, generator=g) * 0.5
    coords = (torch.rand(2, 64, 3, generator=g) * 2.6 - 1.3)  # includes out-of-range points (zeros padding)
    coords[0, :4] = torch.tensor([[1.0, -1.0, 0.0], [0.999, 0.0, -0.999], [-1.0625, 0.5, 0.5], [0.0, 0.0, 0.0]])
    out["g1_planes"] = planes.numpy()
    out["g1_coords"] = coords.numpy()
    out["g1_v1"] = gu.sample_from_planes(planes, coords, interpolate_feat="v1").numpy()
    out["g1_v2"] = gu.sample_from_planes(planes, coords, interpolate_feat="v2").numpy()
    out["g1_proj"] = gu.project_onto_planes(gu.planes, coords).numpy()  # (N*3, M, 2)

    # ---------------- G2: VanillaMLP (imported reference) ------------------------------
    cfg = dict(otype="VanillaMLP", activation="ReLU", output_activation="none", n_neurons=64, n_hidden_layers=2)
    torch.manual_seed(7)
    sdf_net = nw.get_mlp(32, 1, cfg)
    feat_net = nw.get_mlp(96, 3, cfg)
    xs = torch.randn(50, 32, generator=g)
    xf = torch.randn(50, 96, generator=g)
    sdf_w = [sdf_net.layers[i].weight.detach().clone() for i in (0, 2, 4)]
    feat_w = [feat_net.layers[i].weight.detach().clone() for i in (0, 2, 4)]
    for i, w in enumerate(sdf_w):
        out[f"g2_sdf_w{i}"] = w.numpy()
    for i, w in enumerate(feat_w):
        out[f"g2_feat_w{i}"] = w.numpy()
    out["g2_xs"], out["g2_xf"] = xs.numpy(), xf.numpy()
    with torch.no_grad():
        out["g2_ys"] = sdf_net(xs.clone()).numpy()
        out["g2_yf"] = feat_net(xf.clone()).numpy()
        out["g2_rgb"] = nw.get_activation("sigmoid-mipnerf")(feat_net(xf.clone())).numpy()

    # ---------------- G3/G4: rotation + projection composite, sdf + first-order normals ----
    # Reference semantics of few_step_triplane_dual_stable_diffusion.py:198-351 assembled
    # from the imported reference functions + torch.transpose/rot90 (the geometry class
    # itself cannot be imported: threestudio/pytorch_lightning/diffusers are absent).
    cache = torch.randn(2, 6, 32, 16, 16, generator=g) * 0.5
    pts = (torch.rand(2, 40, 3, generator=g) * 2.4 - 1.2).requires_grad_(True)
    rot = torch.zeros_like(cache)
    rot[:, 0::3] = torch.transpose(cache[:, 0::3], 3, 4)
    rot[:, 1::3] = torch.rot90(cache[:, 1::3], k=2, dims=(3, 4))
    rot[:, 2::3] = torch.rot90(cache[:, 2::3], k=-1, dims=(3, 4))
    bbox = torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]])
    p_scaled = gu.contract_to_unisphere_custom(pts, bbox, False)
    enc_geo = gu.sample_from_planes(rot[:, 0:3].contiguous(), p_scaled, interpolate_feat="v1")
    enc_tex = gu.sample_from_planes(rot[:, 3:6].contiguous(), p_scaled, interpolate_feat="v2")
    sdf_orig = sdf_net(enc_geo).view(2, 40, 1)
    sdf = sdf_orig + ((pts ** 2).sum(dim=-1, keepdim=True).sqrt() - 0.5)
    feats = feat_net(enc_tex)
    sdf_grad = torch.autograd.grad(sdf, pts, grad_outputs=torch.ones_like(sdf))[0]
    normal = torch.nn.functional.normalize(sdf_grad, dim=-1)
    out["g3_cache"] = cache.numpy()
    out["g3_pts"] = pts.detach().numpy()
    out["g3_enc_geo"] = enc_geo.detach().numpy()
    out["g3_enc_tex"] = enc_tex.detach().numpy()
    out["g4_sdf"] = sdf.detach().numpy()
    out["g4_features"] = feats.detach().numpy()
    out["g4_sdf_grad"] = sdf_grad.numpy()
    out["g4_normal"] = normal.numpy()
    np.savez_compressed(os.path.join(HERE, "reference_ops.npz"), **out)
    print("wrote reference_ops.npz", {k: v.shape for k, v in out.items()})

    # ---------------- G5/G6: full render + backward (oracle, fp64 and fp32) -----------
    gen = torch.Generator().manual_seed(99)
    P, n_view, Hh, Ww, S, R = 1, 2, 6, 6, 16, 16
    cache64 = (torch.randn(P, 6, 32, R, R, generator=gen, dtype=torch.float64) * 0.5)
    sdf_w64 = O.init_mlp_weights([32, 64, 64, 1], gen, torch.float64)
    feat_w64 = O.init_mlp_weights([96, 64, 64, 3], gen, torch.float64)
    rays_o, rays_d, c2w, cam_d = O.make_cameras(n_view, Hh, Ww, dtype=torch.float32)
    n_rays = n_view * Hh * Ww
    ts, te = O.uniform_intervals(n_rays, S, 0.1, 4.0)
    # make the intervals ragged in width (importance-sampler-like) but sorted
    jitter = torch.rand(n_rays, S + 1, generator=gen) * 0.5
    edges = torch.cat([ts, te[:, -1:]], 1) + jitter * (3.9 / S)
    edges, _ = torch.sort(edges, dim=1)
    ts, te = edges[:, :-1].contiguous(), edges[:, 1:].contiguous()
    bg = torch.tensor([1.0, 1.0, 1.0])
    proj = {k: torch.randn(n_view, Hh, Ww, c, generator=gen) for k, c in
            (("comp_rgb", 3), ("opacity", 1), ("depth", 1), ("z_variance", 1), ("disparity", 1),
             ("comp_normal", 3), ("comp_normal_cam_vis", 3))}
    # the sampled scene must contain a surface: bias weights so sdf crosses zero inside the box
    res = {}
    res.update(cache=cache64.float().numpy(), rays_o=rays_o.numpy(), rays_d=rays_d.numpy(), c2w=c2w.numpy(),
               cam_d=cam_d.numpy(), t_starts=ts.numpy(), t_ends=te.numpy(), bg=bg.numpy())
    for i, w in enumerate(sdf_w64):
        res[f"sdf_w{i}"] = w.float().numpy()
    for i, w in enumerate(feat_w64):
        res[f"feat_w{i}"] = w.float().numpy()
    for k, v in proj.items():
        res[f"proj_{k}"] = v.numpy()
    for tag, dt in (("f64", torch.float64), ("f32", torch.float32)):
        c = cache64.float().to(dt).requires_grad_(True)
        sw = [w.float().to(dt).requires_grad_(True) for w in sdf_w64]
        fw = [w.float().to(dt).requires_grad_(True) for w in feat_w64]
        o = O.render(c, sw, fw, rays_o.to(dt), rays_d.to(dt), ts.to(dt), te.to(dt), bg.to(dt), cam_d.to(dt),
                     c2w.to(dt), inv_std=100.0, rgb_grad_shrink=0.5)
        loss = O.synthetic_loss(o, {k: v.to(dt) for k, v in proj.items()})
        grads = torch.autograd.grad(loss, [c] + sw + fw)
        for k in ("comp_rgb", "opacity", "depth", "z_variance", "disparity", "comp_normal", "comp_normal_cam_vis",
                  "comp_normal_cam_vis_white", "sdf", "alpha", "weights", "features", "sdf_grad", "normal", "trans"):
            res[f"{tag}_{k}"] = o[k].detach().numpy()
        res[f"{tag}_loss"] = loss.detach().numpy()
        res[f"{tag}_g_cache"] = grads[0].numpy()
        for i in range(3):
            res[f"{tag}_g_sdf_w{i}"] = grads[1 + i].numpy()
            res[f"{tag}_g_feat_w{i}"] = grads[4 + i].numpy()
    np.savez_compressed(os.path.join(HERE, "render_small.npz"), **res)
    print("wrote render_small.npz; opacity range", res["f64_opacity"].min(), res["f64_opacity"].max())

    # ---------------- G7: K1 (grad2_2d) known-answer vectors (oracle, fp64) -------------
    gen = torch.Generator().manual_seed(5)
    N, C, H, W, M = 3, 4, 7, 9, 33
    inp = torch.randn(N, C, H, W, generator=gen, dtype=torch.float64)
    grid = torch.rand(N, 1, M, 2, generator=gen, dtype=torch.float64) * 2.4 - 1.2
    go = torch.randn(N, C, 1, M, generator=gen, dtype=torch.float64)
    g2i = torch.randn(N, C, H, W, generator=gen, dtype=torch.float64)
    g2g = torch.randn(N, 1, M, 2, generator=gen, dtype=torch.float64)
    ggo, gi, gg = O.grid_sample_2d_grad2(g2i, g2g, go, inp, grid)
    np.savez_compressed(os.path.join(HERE, "grad2_kat.npz"), inp=inp.numpy(), grid=grid.numpy(), go=go.numpy(),
                        g2i=g2i.numpy(), g2g=g2g.numpy(), ggo=ggo.numpy(), gi=gi.numpy(), gg=gg.numpy())
    print("wrote grad2_kat.npz")


if __name__ == "__main__":
    main()
