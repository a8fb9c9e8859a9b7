"""Batched Woodbury correction (bkcore_woodbury_correct through the C ABI) against outputs of the reference's own
BirmanSchwingerCore.compute_semiseparable_resolvent (tests/golden/woodbury_golden.npz, written by
oracle/gen_golden_woodbury.py in the build container) and against a float64 restatement of its loop."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "woodbury_golden.npz")


def _loop_f64(G, U, V):
    """birman_schwinger_core.py:570-604 in complex128."""
    G = G.to(torch.complex128)
    Uc, Vc = U.to(torch.complex128), V.to(torch.complex128)
    out = torch.zeros_like(G)
    r = U.shape[1]
    for b in range(G.shape[0]):
        g = G[b]
        M = torch.eye(r, dtype=torch.complex128) + (Vc.T * g.unsqueeze(0)) @ Uc
        GUi = (g.unsqueeze(1) * Uc) @ torch.linalg.inv(M)
        out[b] = g - (GUi * Vc).sum(dim=1) * g
    return out


def test_woodbury_golden_consistent_with_restatement():
    """The fixture itself: the reference's corrected diagonal equals the float64 restatement applied to its own
    tridiagonal diagonal (complex64 arithmetic there: 1e-4 of the largest entry)."""
    z = np.load(GOLD, allow_pickle=False)
    for n in [str(x) for x in z["__names__"]]:
        G = torch.from_numpy(z[f"{n}/G_tridiag"])
        want = torch.from_numpy(z[f"{n}/G_corrected"]).to(torch.complex128)
        got = _loop_f64(G, torch.from_numpy(z[f"{n}/U"]), torch.from_numpy(z[f"{n}/V"]))
        assert float((got - want).abs().max()) <= 1e-4 * float(want.abs().max()), n


def test_woodbury_refuses_cpu():
    import bkcore
    with pytest.raises(RuntimeError, match="ROCm device"):
        bkcore.woodbury_correct(torch.zeros(2, 8, dtype=torch.complex64), torch.zeros(8, 3), torch.zeros(8, 3))


@pytest.mark.gpu
def test_woodbury_matches_reference_outputs():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import bkcore
    z = np.load(GOLD, allow_pickle=False)
    for n in [str(x) for x in z["__names__"]]:
        U, V = torch.from_numpy(z[f"{n}/U"]), torch.from_numpy(z[f"{n}/V"])
        want = torch.from_numpy(z[f"{n}/G_corrected"])
        scale = float(want.abs().max())
        # the correction alone, from the reference's own tridiagonal diagonal
        got = bkcore.woodbury_correct(torch.from_numpy(z[f"{n}/G_tridiag"]).cuda(), U.cuda(), V.cuda()).cpu()
        assert float((got - want).abs().max()) <= 1e-4 * scale, n
        # the whole path: potential -> operator -> correction
        zz = complex(z[f"{n}/z"][0], z[f"{n}/z"][1])
        full = bkcore.compute_semiseparable_resolvent(torch.from_numpy(z[f"{n}/v"]).cuda(), zz, U.cuda(), V.cuda()).cpu()
        assert float((full - want).abs().max()) <= 2e-4 * scale, n
        tri = bkcore.compute_semiseparable_resolvent(torch.from_numpy(z[f"{n}/v"]).cuda(), zz).cpu()
        ref_tri = torch.from_numpy(z[f"{n}/G_tridiag"])
        assert float((tri - ref_tri).abs().max()) <= 1e-4 * float(ref_tri.abs().max()), n


@pytest.mark.gpu
# (at least 16 rows and rank <= 16: the matrix-core forms of the two O(N r^2) passes; otherwise the vector forms)
@pytest.mark.parametrize("B,N,r", [(1, 1, 1), (5, 100, 7), (3, 5000, 13), (2, 70000, 17), (130, 300, 22), (4, 64, 0),
                                   (16, 64, 16), (17, 1, 1), (70, 5000, 13), (64, 3001, 16), (33, 1030, 1),
                                   (200, 777, 9), (129, 2049, 8), (40, 4100, 15)])
def test_woodbury_matches_float64_loop(B, N, r):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import bkcore
    g = torch.Generator().manual_seed(B + N + r)
    G = torch.complex(torch.randn(B, N, generator=g), torch.randn(B, N, generator=g)) * 0.4
    amp = 0.5 / max(N, 1) ** 0.25
    U, V = amp * torch.randn(N, r, generator=g), amp * torch.randn(N, r, generator=g)
    want = _loop_f64(G, U, V) if r else G.to(torch.complex128)
    got = bkcore.woodbury_correct(G.cuda(), U.cuda(), V.cuda()).cpu().to(torch.complex128)
    assert float((got - want).abs().max()) <= 1e-4 * float(want.abs().max())


@pytest.mark.gpu
def test_woodbury_singular_row_is_left_uncorrected_and_limits():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import bkcore
    # r = 1, one position: M = 1 + v g u = 0 for g = -1/(u v)
    U, V = torch.tensor([[2.0]]), torch.tensor([[0.5]])
    G = torch.tensor([[-1.0 + 0j], [0.3 + 0.2j]], dtype=torch.complex64)
    out = bkcore.woodbury_correct(G.cuda(), U.cuda(), V.cuda()).cpu()
    assert out[0, 0] == G[0, 0]
    m = 1 + 0.5 * G[1, 0] * 2.0
    assert abs(out[1, 0] - (G[1, 0] - G[1, 0] ** 2 * (2.0 * 0.5) / m)) <= 1e-6
    with pytest.raises(ValueError, match="rank"):
        bkcore.woodbury_correct(torch.zeros(1, 4, dtype=torch.complex64).cuda(), torch.zeros(4, 23).cuda(),
                                torch.zeros(4, 23).cuda())


@pytest.mark.gpu
def test_woodbury_at_the_measured_shape():
    """B = 1024, N = 32768, r = 15 (the shape of the measurements; matrix-core forms): a subset of rows against the
    float64 loop -- rows are independent -- and run-to-run bitwise repeatability."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import bkcore
    B, N, r = 1024, 32768, 15
    g = torch.Generator(device="cuda").manual_seed(5)
    G = 0.4 * torch.complex(torch.randn(B, N, generator=g, device="cuda"), torch.randn(B, N, generator=g, device="cuda"))
    amp = 0.5 / N ** 0.25
    U = amp * torch.randn(N, r, generator=g, device="cuda")
    V = amp * torch.randn(N, r, generator=g, device="cuda")
    out = bkcore.woodbury_correct(G, U, V)
    out2 = bkcore.woodbury_correct(G, U, V)
    assert torch.equal(torch.view_as_real(out), torch.view_as_real(out2))
    rows = [0, 15, 16, 63, 64, 511, 1023]
    want = _loop_f64(G[rows].cpu(), U.cpu(), V.cpu())
    got = out[rows].cpu().to(torch.complex128)
    assert float((got - want).abs().max()) <= 1e-4 * float(want.abs().max())
    # and the same rows as a batch too small for the matrix-core forms: the vector kernels agree to rounding
    small = bkcore.woodbury_correct(G[rows[:3]].contiguous(), U, V).cpu().to(torch.complex128)
    assert float((small - want[:3]).abs().max()) <= 1e-4 * float(want.abs().max())
