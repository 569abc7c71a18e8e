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


@pytest.mark.parametrize("placement", ["tt", "center"])
@pytest.mark.parametrize("n_prop,n_fine,n_rays", [(128, 64, 37), (32, 16, 5), (200, 7, 9), (3, 100, 4), (64, 64, 1)])
def test_importance_sampling_equals_oracle_contract(n_prop, n_fine, n_rays, placement):
    from triplaneturbo_amd import sampler
    step = 1.732 * 2 / 64
    a = sampler.importance_sampling(_sdf_fn, n_rays, n_prop, n_fine, 0.1, 4.0, 100.0, step, device="cuda",
                                    placement=placement)
    b = O.importance_sampling(_sdf_fn, n_rays, n_prop, n_fine, 0.1, 4.0, 100.0, step, placement=placement)
    b64 = O.importance_sampling(_sdf_fn, n_rays, n_prop, n_fine, 0.1, 4.0, 100.0, step, dtype=torch.float64,
                                placement=placement)
    M = n_prop + n_fine + 1  # 129 + 65 edges -> 193 intervals at the reference sizes
    assert a[0].shape == (n_rays, M) and a[1].shape == (n_rays, M)
    ts, te = a[0].cpu(), a[1].cpu()
    assert (te >= ts).all() and (ts[:, 1:] == te[:, :-1]).all()
    if placement == "tt":  # end points pinned
        assert ts[:, 0].eq(0.1).all() and torch.allclose(te[:, -1], torch.tensor(4.0))
    else:  # nothing pinned: the first / last edge sit half a level-0 cell inside
        assert (ts[:, 0] > 0.1).all() and (te[:, -1] < 4.0).all()
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


@pytest.mark.parametrize("placement", ["tt", "center"])
def test_stratified_sampling_equals_oracle_contract_on_the_same_draws(placement):
    """The jittered forms of both placements against the oracle fed with the SAME U[0,1) draws."""
    from triplaneturbo_amd import ops
    g = torch.Generator().manual_seed(8)
    n_rays, K, F, near, far, step = 13, 128, 64, 0.1, 4.0, 1.732 * 2 / 64
    j0, j1 = torch.rand(n_rays, K + 1, generator=g), torch.rand(n_rays, F + 1, generator=g)
    ts, te = ops.sample_uniform(n_rays, K, near, far, "cuda", j0.cuda(), placement=placement)
    wts, wte = O.uniform_intervals(n_rays, K, near, far, placement=placement, jitter=j0)
    torch.testing.assert_close(ts.cpu(), wts, rtol=0, atol=2e-6)
    torch.testing.assert_close(te.cpu(), wte, rtol=0, atol=2e-6)
    a = ops.sample_importance(ts, te, _sdf_fn(ts, te), F, 100.0, step, j1.cuda(), placement=placement)
    b = O.importance_sampling(_sdf_fn, n_rays, K, F, near, far, 100.0, step, placement=placement, jitter0=j0, jitter1=j1)
    b64 = O.importance_sampling(_sdf_fn, n_rays, K, F, near, far, 100.0, step, dtype=torch.float64, placement=placement,
                                jitter0=j0, jitter1=j1)
    for got, w32, w64 in ((a[0].cpu(), b[0], b64[0]), (a[1].cpu(), b[1], b64[1])):
        e_hip, e_cpu = (got.double() - w64).abs(), (w32.double() - w64).abs()
        bad = e_hip > torch.clamp(4 * e_cpu.max(), min=2e-5)
        assert bad.float().mean().item() <= 0.002, (bad.sum().item(), e_hip.max().item(), e_cpu.max().item())
    assert (a[1] >= a[0]).all() and (a[0][:, 1:] == a[1][:, :-1]).all()


@pytest.mark.parametrize("stratified", [False, True])
@pytest.mark.parametrize("placement", ["tt", "center"])
def test_resampled_edges_follow_the_proposal_cdf(placement, stratified):
    """The contract checked against the MATHEMATICS rather than against its own restatement: whatever the placement
    convention, the F + 1 fine edges are the images of one point per cell of [0,1] under the inverse of the proposal
    cdf, so their empirical distribution function equals the cdf within one cell:
        | #{fine edges <= t} / (F + 1)  -  cdf(t) |  <=  1 / (F + 1)   (+ one more cell for "tt", whose n cells are
    1 / F wide and whose jitter may push an edge into the next one) at every proposal edge t.  The cdf itself is built
    here in fp64 from the sdf (proposal density: renderer :289-297, pinned by the golden vectors)."""
    from triplaneturbo_amd import ops
    g = torch.Generator().manual_seed(3)
    n_rays, K, F, near, far, step, inv_std = 29, 128, 64, 0.1, 4.0, 1.732 * 2 / 64, 100.0
    ts, te = O.uniform_intervals(n_rays, K, near, far)
    sdf = _sdf_fn(ts, te) + 0.05 * torch.randn(n_rays, 1, generator=g)  # a different surface position per ray
    u = torch.rand(n_rays, F + 1, generator=g).cuda() if stratified else None
    o_ts, o_te = ops.sample_importance(ts.cuda(), te.cuda(), sdf.cuda(), F, inv_std, step, u, placement=placement)
    edges = torch.cat([o_ts, o_te[:, -1:]], dim=1).cpu().double()  # K + F + 2 merged edges
    prop = torch.cat([ts, te[:, -1:]], dim=1).double()              # K + 1 proposal edges
    # remove the proposal edges from the merged list (multiset difference: both lists are sorted)
    fine = []
    for r in range(n_rays):
        e, p = edges[r].tolist(), prop[r].float().double().tolist()
        out, i = [], 0
        for v in e:
            if i < len(p) and v == p[i]:
                i += 1
            else:
                out.append(v)
        assert i == len(p) and len(out) == F + 1, (i, len(out))
        fine.append(out)
    fine = torch.tensor(fine, dtype=torch.float64)
    sigma = O.proposal_density(sdf.double(), inv_std, step)
    sd = sigma * (te - ts).double()
    excl = torch.cumsum(torch.cat([torch.zeros_like(sd[:, :1]), sd[:, :-1]], dim=1), dim=1)
    cdf = 1.0 - torch.cat([torch.exp(-excl), torch.zeros_like(sd[:, :1])], dim=1)  # at the K + 1 proposal edges
    ecdf = (fine[:, None, :] <= prop[:, :, None] + 1e-6).double().sum(-1) / (F + 1)
    tol = (1.0 if placement == "center" else 2.0) / (F + 1) + 1.0 / F * (placement == "tt") + 1e-4
    assert (ecdf - cdf).abs().max().item() <= tol, ((ecdf - cdf).abs().max().item(), tol)
    # on average: within half a cell ("center"); "tt" pins an edge to u = 0, which offsets the count by one (1.5 cells)
    assert (ecdf - cdf).abs().mean().item() <= (0.5 if placement == "center" else 1.5) / (F + 1)
