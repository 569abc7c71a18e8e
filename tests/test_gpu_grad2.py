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
        gsg.grad2_2d(*args, 1, False)  # border padding: not built


def test_grid_sample_2d_double_backward_matches_oracle():
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
        loss = (gg ** 2).sum() + out.sum()
        return torch.autograd.grad(loss, (inp, grid))

    gi, gg = second_order(lambda a, b: gsg.grid_sample_2d(a, b, "zeros", False), inp.cuda(), grid.cuda(), w.cuda())

    def oracle_sample(a, b):
        o = O.grid_sample_gather(a, b.reshape(a.shape[0], -1, 2))
        return o.permute(0, 2, 1).reshape(a.shape[0], a.shape[1], 1, -1)

    wi, wg = second_order(oracle_sample, inp.double(), grid.double(), w.double())
    assert (gi.cpu().double() - wi).norm() / wi.norm() < 1e-4
    assert (gg.cpu().double() - wg).norm() / wg.norm() < 1e-4
