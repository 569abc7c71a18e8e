"""The differentiable render step can be captured in a hipGraph (torch.cuda.graph) and replayed: the per-XCD work-queue
counters of the decode kernels are zeroed by a one-wave kernel node in front of each kernel node (a slot of their own per
captured launch), and the composite uses no host-synchronising op.  Replays must reproduce the eager result every
time -- a dirty counter would make a replay pop no work and leave stale outputs.  Also: a dirty counter slot (what a
faulted / killed kernel leaves behind) must not affect the next launch, and launches on different streams must not
share counters."""
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


def _render_setup(seed=23):
    from triplaneturbo_amd import ops
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(seed)
    P, R, Hh, Ww, S = 1, 64, 32, 32, 64
    cache = (torch.randn(P, 6, 32, R, R, generator=g) * 0.5).to(dev)
    sw = [w.to(dev) for w in O.init_mlp_weights([32, 64, 64, 1], g)]
    fw = [w.to(dev) for w in O.init_mlp_weights([96, 64, 64, 3], g)]
    ro, rd, c2w, cd = [t.to(dev) for t in O.make_cameras(1, Hh, Ww)]
    ts, te = [t.to(dev) for t in O.uniform_intervals(Hh * Ww, S, 0.3, 3.2)]
    packed = ops.planes_pack(cache)

    def render():
        return ops.render_forward_raw(packed, sw, fw, ro.reshape(-1, 3), rd.reshape(-1, 3), ts, te, Hh * Ww,
                                      ops.RenderConfig(), image_w=Ww)
    return ops, render


def test_dirty_queue_slot_does_not_affect_the_next_launch():
    import ctypes
    from triplaneturbo_amd import _lib
    ops, render = _render_setup()
    want = render()
    torch.cuda.synchronize()
    for _ in range(3):
        st = _lib.load().tt_debug_poison_queue(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert st == 0
        got = render()
        torch.cuda.synchronize()
        for k in ("opacity", "rgb_fg", "sdf", "features", "weights"):
            assert torch.equal(got[k], want[k]), k  # every item popped exactly once: bit-identical per-sample results


def test_concurrent_streams_do_not_share_queue_counters():
    ops, render = _render_setup(29)
    want = render()
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(3)]
    outs = []
    for rep in range(4):
        for s in streams:
            with torch.cuda.stream(s):
                outs.append(render())
    torch.cuda.synchronize()
    for got in outs:
        for k in ("opacity", "rgb_fg", "sdf", "features"):
            assert torch.equal(got[k], want[k]), k
