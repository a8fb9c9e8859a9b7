"""bkcore_cumsum through the C ABI (fused_associative_scan mirror) against a float64 cumulative sum; shapes of the
reference's AR-SSM layer (B, L, rank) along dim 1 (src/models/phase1/ar_ssm_layer.py:517-525) and the scan's own
test shapes (src/kernels/associative_scan.py:311-330: (4, L, 512) along the last dim... any dim)."""
import numpy as np
import pytest
import torch


def test_cumsum_refuses_cpu_and_bad_args():
    import bkcore
    with pytest.raises(RuntimeError, match="ROCm device"):
        bkcore.fused_associative_scan(torch.zeros(4, 4))
    with pytest.raises(IndexError):
        bkcore.fused_associative_scan(torch.zeros(4, 4), dim=2)


CASES = [((4, 512, 32), 1), ((8, 2048, 32), 1), ((2, 100, 7), 1), ((3, 8192), -1), ((5, 1000), 1), ((2, 3, 65, 33), 2),
         ((1, 1, 1), 1), ((6, 1, 40), 1), ((4, 777, 64), 0), ((2, 64, 129), 1), ((1, 70000), -1),
         ((3, 4099), -1), ((2, 1001), -1), ((2, 5000), -1)]   # last dim: rows not 16-byte aligned, ragged passes


@pytest.mark.gpu
@pytest.mark.parametrize("shape,dim", CASES)
@pytest.mark.parametrize("reverse", [False, True])
def test_cumsum_matches_float64(shape, dim, reverse):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import bkcore
    g = torch.Generator().manual_seed(sum(shape) + dim)
    x = torch.randn(*shape, generator=g)
    xd = x.double()
    want = torch.flip(torch.cumsum(torch.flip(xd, [dim]), dim), [dim]) if reverse else torch.cumsum(xd, dim)
    got = bkcore.fused_associative_scan(x.cuda(), dim=dim, reverse=reverse)
    assert got.shape == x.shape and got.dtype == torch.float32
    L = shape[dim]
    # float32 running sums of L standard normals: error of a few ulp of sqrt(L)
    tol = 4e-7 * max(L, 4) ** 0.5 * max(1.0, float(want.abs().max()))
    assert float((got.cpu().double() - want).abs().max()) <= tol
    # against the tensor library's own float32 result as well (different summation order)
    lib = torch.flip(torch.cumsum(torch.flip(x.cuda(), [dim]), dim), [dim]) if reverse else torch.cumsum(x.cuda(), dim)
    assert float((got - lib).abs().max()) <= 2 * tol


@pytest.mark.gpu
def test_cumsum_integers_are_exact_and_gradient_is_reverse_scan():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import bkcore
    g = torch.Generator().manual_seed(3)
    xi = torch.randint(-8, 9, (3, 5000, 32), generator=g).float()
    got = bkcore.fused_associative_scan(xi.cuda(), dim=1)
    assert torch.equal(got.cpu(), torch.cumsum(xi, 1))                  # small integers: every partial sum is exact
    x = torch.randn(2, 300, 16, generator=g).cuda().requires_grad_(True)
    w = torch.randn(2, 300, 16, generator=g).cuda()
    (bkcore.fused_associative_scan(x, dim=1) * w).sum().backward()
    want = torch.flip(torch.cumsum(torch.flip(w.double(), [1]), 1), [1])
    assert float((x.grad.double() - want).abs().max()) <= 1e-4
    xb = torch.randn(2, 64, 8, generator=g).to(torch.bfloat16).cuda()
    yb = bkcore.fused_associative_scan(xb, dim=1)
    assert yb.dtype == torch.bfloat16
    assert float((yb.float() - torch.cumsum(xb.float(), 1)).abs().max()) <= 0.1


@pytest.mark.gpu
@pytest.mark.parametrize("reverse", [False, True])
def test_cumsum_last_dim_infinity_and_mixed_magnitudes(reverse):
    """Row kernel: an infinity poisons only the positions from it onwards (as torch.cumsum), and a huge group does not
    cancel the prefix that came before it."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import bkcore
    g = torch.Generator().manual_seed(11)
    x = torch.randn(3, 3000, generator=g)
    x[0, 1501] = float("inf")
    x[1, 700] = 3.0e7          # partial sums before it are O(30): lost if the prefix were formed by subtraction
    x[1, 701] = -3.0e7
    xd = x.double()
    want = torch.flip(torch.cumsum(torch.flip(xd, [1]), 1), [1]) if reverse else torch.cumsum(xd, 1)
    got = bkcore.fused_associative_scan(x.cuda(), dim=1, reverse=reverse).cpu().double()
    fin = torch.isfinite(want)
    assert torch.equal(torch.isfinite(got), fin)
    assert torch.equal(got[~fin], want[~fin])
    # row 1 away from the two huge entries: float32 accuracy of an O(sqrt(L)) running sum, a few units at 3e7's ulp
    # only where the huge entry is part of the sum
    sel = torch.ones_like(fin)
    sel[1, 700:702] = False
    assert float((got - want)[fin & sel].abs().max()) <= 4.0
    sel[1] = False
    assert float((got - want)[fin & sel].abs().max()) <= 1e-3
