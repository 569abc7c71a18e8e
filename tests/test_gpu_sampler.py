"""The HIP samplers (tt_sample_uniform, tt_sample_importance) against the oracle's sampling contract
(oracle/cpu_ref.py::uniform_intervals / importance_sampling; the nerfacc boundary, parity unpinned).  Edge values
must agree to fp32 rounding of the transmittance scan; structure (sortedness, shared edges, end points) exactly."""
import pytest
import torch

from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu


def _sdf_fn(ts, te):  # any deterministic smooth function of the mid-points
    tm = (ts + te) / 2
    return 0.4 * torch.cos(3.0 * tm) + 0.1 * (tm - 1.5)


@pytest.mark.parametrize("n_prop,n_fine,n_rays", [(128, 64, 37), (32, 16, 5), (200, 7, 9), (3, 100, 4), (64, 64, 1)])
def test_importance_sampling_equals_oracle_contract(n_prop, n_fine, n_rays):
    from triplaneturbo_amd import sampler
    step = 1.732 * 2 / 64
    a = sampler.importance_sampling(_sdf_fn, n_rays, n_prop, n_fine, 0.1, 4.0, 100.0, step, device="cuda")
    b = O.importance_sampling(_sdf_fn, n_rays, n_prop, n_fine, 0.1, 4.0, 100.0, step)
    b64 = O.importance_sampling(_sdf_fn, n_rays, n_prop, n_fine, 0.1, 4.0, 100.0, step, dtype=torch.float64)
    M = n_prop + n_fine + 1  # 129 + 65 edges -> 193 intervals at the reference sizes
    assert a[0].shape == (n_rays, M) and a[1].shape == (n_rays, M)
    ts, te = a[0].cpu(), a[1].cpu()
    assert (te >= ts).all() and (ts[:, 1:] == te[:, :-1]).all()
    assert ts[:, 0].eq(0.1).all() and torch.allclose(te[:, -1], torch.tensor(4.0))
    # as close to the fp64 contract as the fp32 oracle is (x4), or 2e-5 (the cdf is flat to fp32 behind the surface:
    # there the inverse is ill-conditioned for every fp32 implementation)
    for got, w32, w64 in ((ts, b[0], b64[0]), (te, b[1], b64[1])):
        e_hip = (got.double() - w64).abs()
        e_cpu = (w32.double() - w64).abs()
        bad = e_hip > torch.clamp(4 * e_cpu.max(), min=2e-5)
        assert bad.float().mean().item() <= 0.002, (bad.sum().item(), e_hip.max().item(), e_cpu.max().item())


def test_uniform_and_stratified_intervals():
    from triplaneturbo_amd import sampler
    ts, te = sampler.uniform_intervals(7, 128, 0.1, 4.0, device="cuda")
    wts, wte = O.uniform_intervals(7, 128, 0.1, 4.0)
    torch.testing.assert_close(ts.cpu(), wts, rtol=0, atol=0)
    torch.testing.assert_close(te.cpu(), wte, rtol=0, atol=0)
    g = torch.Generator(device="cuda").manual_seed(0)
    ts, te = sampler.uniform_intervals(5, 32, 0.1, 4.0, device="cuda", stratified=True, generator=g)
    ts, te = ts.cpu(), te.cpu()
    assert (te > ts).all() and ts[:, 0].eq(0.1).all() and torch.allclose(te[:, -1], torch.tensor(4.0))
    assert (ts[:, 1:] == te[:, :-1]).all()
    cell = 3.9 / 32
    edges = 0.1 + cell * torch.arange(1, 32)
    assert ((ts[:, 1:] - edges).abs() <= 0.5 * cell + 1e-6).all()  # every interior edge stays in its own cell
    assert ts.std(dim=0)[1:].min() > 0  # and is actually jittered per ray


def test_stratified_importance_is_a_pure_function_of_its_random_inputs():
    from triplaneturbo_amd import ops
    g = torch.Generator().manual_seed(4)
    n_rays, K, F = 11, 128, 64
    ts, te = O.uniform_intervals(n_rays, K, 0.1, 4.0)
    sdf = _sdf_fn(ts, te)
    u = torch.rand(n_rays, F + 1, generator=g)
    a = ops.sample_importance(ts.cuda(), te.cuda(), sdf.cuda(), F, 100.0, 0.054, u.cuda())
    b = ops.sample_importance(ts.cuda(), te.cuda(), sdf.cuda(), F, 100.0, 0.054, u.cuda())
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    # the jittered fine edges are a different set from the deterministic ones, still sorted and covering [near, far]
    c = ops.sample_importance(ts.cuda(), te.cuda(), sdf.cuda(), F, 100.0, 0.054, None)
    assert not torch.equal(a[0], c[0])
    assert (a[0][:, 1:] >= a[0][:, :-1]).all() and a[0][:, 0].eq(0.1).all()
