"""The differentiable render step can be captured in a hipGraph (torch.cuda.graph) and replayed: the per-XCD work-queue
counters of the decode kernels reset themselves (the last wave of a launch zeroes its slot), and the composite uses no
host-synchronising op.  Replays must reproduce the eager result every time -- a dirty counter would make a replay pop
no work and leave stale outputs."""
import pytest
import torch

from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu


def test_captured_step_replays_like_eager():
    from triplaneturbo_amd import functional, ops
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(17)
    P, R, Hh, Ww, S = 1, 32, 16, 16, 48
    cache = (torch.randn(P, 6, 32, R, R, generator=g) * 0.5).to(dev).requires_grad_(True)
    sw = [w.to(dev).requires_grad_(True) for w in O.init_mlp_weights([32, 64, 64, 1], g)]
    fw = [w.to(dev).requires_grad_(True) for w in O.init_mlp_weights([96, 64, 64, 3], g)]
    ro, rd, c2w, cd = [t.to(dev) for t in O.make_cameras(1, Hh, Ww)]
    ts, te = [t.to(dev) for t in O.uniform_intervals(Hh * Ww, S, 0.3, 3.2)]
    bg = torch.ones(3, device=dev)
    rc = ops.RenderConfig(inv_std=50.0)
    params = [cache] + sw + fw
    static = [torch.zeros_like(p) for p in params]

    def step():
        out = functional.volume_render(cache, sw, fw, ro, rd, ts, te, bg, cd, c2w, rc, training=True)
        loss = out["comp_rgb"].mean() + out["comp_normal_cam_vis"].mean() + \
            ((out["sdf_grad"].norm(dim=-1) - 1) ** 2).mean()
        for dst, gr in zip(static, torch.autograd.grad(loss, params)):
            dst.copy_(gr)
        return loss

    step()  # first launch allocates the library's counter scratch: must happen outside a capture
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    eager_loss = step().item()
    eager = [t.clone() for t in static]
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        loss = step()
    for _ in range(4):
        for t in static:
            t.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert abs(loss.item() - eager_loss) <= 1e-5 * abs(eager_loss)
        for a, b in zip(static, eager):  # plane gradients accumulate with atomics: equal up to summation order
            assert ((a - b).norm() / b.norm()).item() < 1e-5
