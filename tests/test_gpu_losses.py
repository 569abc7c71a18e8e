"""The fused eikonal regulariser (tt_eikonal_fwd / _bwd) against the torch expression the reference's training loop
uses (multiprompt_dual_renderer_multistep_generator.py:696-699), values and gradients, incl. zero-length rows."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 257, 100003])
def test_eikonal_loss_matches_torch(n):
    from triplaneturbo_amd import ops
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, 3, generator=g) * 1.3
    if n > 4:
        x[3] = 0.0  # ||g|| = 0: torch's norm backward gives 0 there
    xd = x.double().requires_grad_(True)
    want = ((torch.linalg.norm(xd, ord=2, dim=-1) - 1.0) ** 2).mean() * 0.7
    gw, = torch.autograd.grad(want, [xd])
    xg = x.cuda().requires_grad_(True)
    got = ops.eikonal_loss(xg) * 0.7
    gg, = torch.autograd.grad(got, [xg])
    torch.testing.assert_close(got.cpu().double(), want.detach(), rtol=2e-6, atol=1e-7)
    torch.testing.assert_close(gg.cpu().double(), gw, rtol=2e-5, atol=1e-9)


def test_eikonal_loss_is_bit_reproducible_and_survives_huge_values():
    """The forward combines its per-block partial sums in a fixed-point integer accumulator: repeated launches are
    BIT-identical whatever order the blocks arrive in; a diverged field (block share of the mean >= 1024) takes the
    float fallback and is still right."""
    from triplaneturbo_amd import ops
    g = torch.Generator().manual_seed(7)
    x = (torch.randn(300007, 3, generator=g) * 1.1).cuda()
    ref = ops.eikonal_loss(x).item()
    for _ in range(20):
        assert ops.eikonal_loss(x).item() == ref
    big = (torch.randn(70001, 3, generator=g) * 4.0e3).cuda()
    want = ((torch.linalg.norm(big.double(), ord=2, dim=-1) - 1.0) ** 2).mean().item()
    assert abs(ops.eikonal_loss(big).item() - want) <= 2e-6 * want
    assert torch.isnan(ops.eikonal_loss(torch.full((5, 3), float("nan")).cuda()))


def test_eikonal_loss_on_a_misaligned_view():
    """tt_eikonal_fwd reads three float4 per four samples when the pointer is 16-byte aligned and falls back to scalar
    loads otherwise: a view starting one sample (12 bytes) into a buffer must give the same value as its aligned copy."""
    from triplaneturbo_amd import ops
    g = torch.Generator().manual_seed(11)
    buf = (torch.randn(50003, 3, generator=g) * 1.2).cuda()
    view = buf[1:]
    assert view.data_ptr() % 16 != 0 and view.is_contiguous()
    a, b = ops.eikonal_loss(view).item(), ops.eikonal_loss(view.clone()).item()
    want = ((torch.linalg.norm(view.double(), ord=2, dim=-1) - 1.0) ** 2).mean().item()
    assert abs(a - want) <= 2e-6 * want and abs(b - want) <= 2e-6 * want
    assert abs(a - b) <= 1e-6 * want  # same terms, different per-thread grouping
