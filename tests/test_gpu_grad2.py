"""GPU test of the operator-level drop-in for the reference's native op (gridsample_cuda.cu grad2_2d)."""
import os

import numpy as np
import pytest
import torch

from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_grad2_2d_known_answers(golden_dir):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from triplaneturbo_amd import grid_sample_gradfix as gsg
    k = dict(np.load(os.path.join(golden_dir, "grad2_kat.npz")))
    args = [T(k[n]).float().cuda() for n in ("g2i", "g2g", "go", "inp", "grid")]
    ggo, gi, gg = gsg.grad2_2d(*args, 0, False)
    for got, name in ((ggo, "ggo"), (gi, "gi"), (gg, "gg")):
        want = T(k[name])
        err = (got.cpu().double() - want).abs().max().item()
        assert err <= 2e-5 * max(1.0, want.abs().max().item()), (name, err)
    with pytest.raises(RuntimeError, match="unsupported"):
        gsg.grad2_2d(*args, 2, False)  # reflection padding: the reference's op takes a bool, never built


@pytest.mark.parametrize("padding_mode,align_corners,dtype", [
    ("zeros", False, torch.float32), ("zeros", True, torch.float32), ("border", False, torch.float32),
    ("border", True, torch.float32), ("zeros", False, torch.float64), ("border", True, torch.float64)])
def test_grid_sample_2d_double_backward_matches_oracle(padding_mode, align_corners, dtype):
    """The float / double cases of the reference's dispatch (gridsample_cuda.cu:560-594; zeros / border, either
    align_corners) against autograd through the gather-based oracle in fp64 (itself checked against torch's
    grid_sample forward + first backward for all four modes, tests/test_oracle_golden.py).  Half: next test."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from triplaneturbo_amd import grid_sample_gradfix as gsg
    g = torch.Generator().manual_seed(0)
    inp = torch.randn(3, 8, 16, 16, generator=g)
    grid = torch.rand(3, 1, 200, 2, generator=g) * 2.4 - 1.2
    w = torch.randn(3, 8, 1, 200, generator=g)

    def second_order(f_sample, inp, grid, w):
        inp = inp.clone().requires_grad_(True)
        grid = grid.clone().requires_grad_(True)
        out = f_sample(inp, grid)
        (gg,) = torch.autograd.grad((out * w).sum(), grid, create_graph=True)  # like the analytic normal
        loss = (gg.to(torch.float64) ** 2).sum() + out.to(torch.float64).sum()  # (a half sum would overflow)
        return torch.autograd.grad(loss, (inp, grid))

    gi, gg = second_order(lambda a, b: gsg.grid_sample_2d(a, b, padding_mode, align_corners), inp.cuda().to(dtype),
                          grid.cuda().to(dtype), w.cuda().to(dtype))

    def oracle_sample(a, b):
        o = O.grid_sample_gather(a, b.reshape(a.shape[0], -1, 2), padding_mode, align_corners)
        return o.permute(0, 2, 1).reshape(a.shape[0], a.shape[1], 1, -1)

    wi, wg = second_order(oracle_sample, inp.double(), grid.double(), w.double())
    tol = {torch.float32: 1e-4, torch.float64: 1e-12}[dtype]
    assert (gi.cpu().double() - wi).norm() / wi.norm() < tol
    assert (gg.cpu().double() - wg).norm() / wg.norm() < tol


@pytest.mark.parametrize("padding_mode,align_corners", [(0, False), (1, True)])
def test_grad2_2d_half_matches_its_float_path(padding_mode, align_corners):
    """Half (gridsample_cuda.cu:560 dispatches it): same operands rounded to half, the op in half vs the op in fp32.
    Half I/O, fp32 arithmetic inside; grad_input is accumulated with packed-half atomics (every add rounds to 11 bits)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from triplaneturbo_amd import grid_sample_gradfix as gsg
    g = torch.Generator().manual_seed(3)
    N, C, H, W, M = 2, 8, 16, 16, 300
    ops = [torch.randn(N, C, H, W, generator=g), torch.randn(N, 1, M, 2, generator=g), torch.randn(N, C, 1, M, generator=g),
           torch.randn(N, C, H, W, generator=g), torch.rand(N, 1, M, 2, generator=g) * 2.4 - 1.2]
    h = [t.half().cuda() for t in ops]
    f = [t.float() for t in h]
    got = gsg.grad2_2d(*h, padding_mode, align_corners)
    want = gsg.grad2_2d(*f, padding_mode, align_corners)
    for a, b, tol in zip(got, want, (2e-3, 2e-2, 2e-3)):
        assert a.dtype == torch.float16
        assert ((a.float() - b).norm() / b.norm()).item() < tol
