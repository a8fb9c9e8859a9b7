"""Whole rows (one workgroup per row) whose chunks at the right end fail their checks are only staged from there on and
go to the scan kernel; which rows those are depends on each row's own data alone.  The reference benchmark's inputs
(randn a, b, c; scripts/benchmark_bk_triton.py:66-70) are of that kind, every row of them.
Checked here: results equal those of the scan kernel alone, bit for bit; the other rows of the batch keep their bits;
the same call twice gives the same bits whatever ran before; and the operator's NaN rules (bk_core.py:256-264), which
concern the whole batch, still see a NaN in the part of a row that was only staged."""
import numpy as np
import pytest
import torch

from conftest import max_rel_err
from oracle import bk_oracle_c as C

pytestmark = pytest.mark.gpu
CH = 4096


@pytest.fixture()
def bk():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import bkcore
    from bkcore import bk_core
    yield bkcore
    bk_core._DEBUG["keep_ws"] = False
    bk_core._DEBUG["last_ws"] = None
    bk_core._DEBUG["scan_path"] = False


def _good(B, N, seed):
    g = torch.Generator().manual_seed(seed)
    a = -2.0 + 0.5 * torch.randn(B, N, generator=g)
    return a.cuda(), torch.ones(B, N - 1).cuda(), torch.ones(B, N - 1).cuda()


def _fwd(bk, a, b, c, z):
    from bkcore import bk_core
    bk_core._DEBUG["keep_ws"] = True
    try:
        _, G = bk.bk_forward(a, b, c, z)
        torch.cuda.synchronize()
        return G, bk_core.rows_recomputed()
    finally:
        bk_core._DEBUG["keep_ws"] = False
        bk_core._DEBUG["last_ws"] = None


def test_rows_lost_at_their_right_end(bk):
    B, N = 300, 8 * CH          # whole rows, eight chunks each
    a, b, c = _good(B, N, seed=3)
    z = 1j
    clean, r0 = _fwd(bk, a, b, c, z)
    assert r0 == 0
    bad_rows = [7, 150, 299]
    for row in bad_rows:        # a stretch in the rightmost chunk that the 16-step warm-up cannot forget its start over
        a[row, N - 900:N - 600] = 0.0
        b[row, N - 900:N - 600] = 3.0
        c[row, N - 900:N - 600] = 3.0
    G, r1 = _fwd(bk, a, b, c, z)
    assert r1 == len(bad_rows)
    keep = [r for r in range(B) if r not in bad_rows]
    assert torch.equal(torch.view_as_real(G[keep]), torch.view_as_real(clean[keep]))
    # the lost rows are the scan kernel's, bit for bit
    from bkcore import bk_core
    bk_core._DEBUG["scan_path"] = True
    try:
        Gs, _ = _fwd(bk, a[bad_rows].contiguous(), b[bad_rows].contiguous(), c[bad_rows].contiguous(), z)
    finally:
        bk_core._DEBUG["scan_path"] = False
    assert torch.equal(torch.view_as_real(Gs), torch.view_as_real(G[bad_rows]))
    ref, _ = C.forward_f32(a[bad_rows].cpu().numpy(), b[bad_rows].cpu().numpy(), c[bad_rows].cpu().numpy(), z)
    got = G[bad_rows].cpu().numpy()
    assert max_rel_err(got, ref[..., 0] + 1j * ref[..., 1]) <= 1e-4
    assert max_rel_err(Gs.cpu().numpy(), ref[..., 0] + 1j * ref[..., 1]) <= 1e-4
    # and again, after a call on other data: the same bits
    _fwd(bk, *_good(64, 4 * CH, seed=9), z)
    G2, _ = _fwd(bk, a, b, c, z)
    assert torch.equal(torch.view_as_real(G2), torch.view_as_real(G))


def test_reference_benchmark_distribution_is_repeatable(bk):
    g = torch.Generator(device="cuda").manual_seed(0)
    a, b, c = (torch.randn(96, n, generator=g, device="cuda") for n in (5 * CH + 17, 5 * CH + 16, 5 * CH + 16))
    z = torch.tensor(0.1 + 0.1j, dtype=torch.complex64, device="cuda")
    G1, r1 = _fwd(bk, a, b, c, z)
    assert r1 == 96
    _fwd(bk, *_good(32, 3 * CH, seed=1), 1j)
    G2, r2 = _fwd(bk, a, b, c, z)
    assert r2 == 96 and torch.equal(torch.view_as_real(G1), torch.view_as_real(G2))
    assert bool(torch.isfinite(torch.view_as_real(G1)).all())


def test_nan_in_the_part_that_was_only_staged(bk):
    """Row 0 is lost in its rightmost chunk and has a NaN in he_diag near its LEFT end, which the first launch then only
    stages; row 5 has a NaN in h0_super.  The sanitiser replaces NaNs in b and c only if he_diag holds one anywhere in
    the batch (bk_core.py:260-264), so row 5 comes out finite only if the staged part of row 0 was looked at."""
    B, N = 40, 6 * CH + 5
    a, b, c = _good(B, N, seed=11)
    z = 1j
    a[0, N - 500:N - 300] = 0.0
    b[0, N - 500:N - 300] = 3.0
    c[0, N - 500:N - 300] = 3.0
    a[0, 100] = float("nan")
    b[5, 2 * CH + 3] = float("nan")
    G, _ = _fwd(bk, a, b, c, z)
    assert bool(torch.isfinite(torch.view_as_real(G)).all())
    ref, _ = C.forward_f32(a.cpu().numpy(), b.cpu().numpy(), c.cpu().numpy(), z)
    assert max_rel_err(G.cpu().numpy(), ref[..., 0] + 1j * ref[..., 1]) <= 1e-4
    # without the NaN in he_diag the NaN of h0_super stays and poisons row 5 (non-finite -> 0 in the output)
    a[0, 100] = -2.0
    G, _ = _fwd(bk, a, b, c, z)
    ref, _ = C.forward_f32(a.cpu().numpy(), b.cpu().numpy(), c.cpu().numpy(), z)
    ref = ref[..., 0] + 1j * ref[..., 1]
    got = G.cpu().numpy()
    others = [r for r in range(B) if r != 5]
    assert max_rel_err(got[others], ref[others]) <= 1e-4
    assert np.all(ref[5] == 0) and np.all(got[5] == 0)
