"""The float64 adjoint formulation of the analytic gradient (oracle/bk_analytic_adjoint.py, the specification of the
HIP analytic backward) against (1) the gradients torch.autograd produced through the reference's own sequential
forward, recorded in the golden fixtures, and (2) autograd through the oracle on seeded cases, both phi index
conventions.  Error measure for gradients: max |x - ref| / max |ref| over the tensor."""
import numpy as np
import pytest
import torch

from oracle import bk_oracle_torch as O
from oracle.bk_analytic_adjoint import adjoint_grads
from test_oracle_golden import ALL

FINITE = [n for n in ALL if n not in ("poisoned", "nan_offdiag_only")]


def rel(x, ref):
    ref = np.asarray(ref)
    if ref.size == 0:
        return 0.0
    return float(np.abs(np.asarray(x) - ref).max() / max(np.abs(ref).max(), 1e-300))


@pytest.mark.parametrize("name", FINITE)
def test_adjoint_matches_reference_autograd(golden, name):
    cs = golden[name]
    a = np.clip(cs["a"].astype(np.float64), -100, 100)
    b = np.clip(cs["b"].astype(np.float64), -10, 10)
    c = np.clip(cs["c"].astype(np.float64), -10, 10)
    gf = cs["gfeat"].astype(np.float64)
    ga, gb, gc, G = adjoint_grads(a, b, c, complex(*cs["z"]), gf[..., 0] + 1j * gf[..., 1], "reference")
    assert rel(G, cs["G"].astype(np.complex128)) <= 2e-7          # complex64 rounding of the fixture
    assert rel(ga, cs["grad_true_a"]) <= 1e-6
    assert rel(gb, cs["grad_true_b"]) <= 1e-6
    assert rel(gc, cs["grad_true_c"]) <= 1e-6


@pytest.mark.parametrize("conv", ["reference", "exact"])
@pytest.mark.parametrize("B,N,z", [(2, 1, 1j), (2, 2, 1j), (3, 5, 0.5j), (2, 300, 0.3 + 0.5j), (1, 1500, 1j)])
def test_adjoint_matches_oracle_autograd(B, N, z, conv):
    g = torch.Generator().manual_seed(100 + N)
    a = -2.0 + 0.5 * torch.randn(B, N, generator=g, dtype=torch.float64)
    b = 1.0 + 0.1 * torch.randn(B, N - 1, generator=g, dtype=torch.float64)
    c = 1.0 + 0.1 * torch.randn(B, N - 1, generator=g, dtype=torch.float64)
    gf = torch.randn(B, N, 2, generator=g, dtype=torch.float64)
    gG = torch.complex(torch.randn(B, N, generator=g, dtype=torch.float64),
                       torch.randn(B, N, generator=g, dtype=torch.float64))
    gbar = (gf[..., 0] + 1j * gf[..., 1] + gG).numpy()
    ga, gb, gc, G = adjoint_grads(a.numpy(), b.numpy(), c.numpy(), z, gbar, conv)
    ta, tb, tc = O.analytic_grads(a, b, c, z, grad_features=gf, grad_G=gG, phi_k_index=conv)
    assert rel(G, O.get_diag(a, b, c, z, phi_k_index=conv, out_dtype=torch.complex128).numpy()) <= 1e-10
    assert rel(ga, ta.numpy()) <= 1e-9
    assert rel(gb, tb.numpy()) <= 1e-9
    assert rel(gc, tc.numpy()) <= 1e-9


def test_adjoint_through_the_magnitude_clip():
    """|G| > 50 somewhere: the clip's derivative (only the phase moves) is part of the gradient."""
    a = np.array([[0.0, 0.3, -0.2, 0.1]])
    b = np.full((1, 3), 0.05)
    c = np.full((1, 3), 0.05)
    z = 0.005j
    gbar = np.array([[1.0 + 0.5j, -0.3 + 1j, 0.7 - 0.2j, 0.1 + 0.1j]])
    ga, gb, gc, G = adjoint_grads(a, b, c, z, gbar, "reference")
    assert np.abs(G).max() == pytest.approx(50.0, rel=1e-9)
    gf = torch.from_numpy(np.stack([gbar.real, gbar.imag], -1))
    ta, tb, tc = O.analytic_grads(torch.from_numpy(a), torch.from_numpy(b), torch.from_numpy(c), z, grad_features=gf)
    assert rel(ga, ta.numpy()) <= 1e-8
    assert rel(gb, tb.numpy()) <= 1e-8
    assert rel(gc, tc.numpy()) <= 1e-8
