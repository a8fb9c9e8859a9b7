"""The operator as the reference's other callers invoke it -- argument kinds, not new arithmetic:
  * NonHermitianBKCore (src/models/phase2/non_hermitian.py:201-245): he_diag is the ``.real`` view of a complex tensor
    (inner stride 2), the off-diagonals are ``expand()``ed nn.Parameters, z is a complex nn.Parameter, and a fifth
    positional argument (use_triton) is passed;
  * PhysicsInformedBKLayer / SparseBKCore / LearnedSparsityG_ii (physics_informed_layer.py:152-163,
    sparse_bk_core.py:316-334, learned_sparsity_g_ii.py:153): four arguments, ``expand()``ed buffers, z a complex64
    buffer;
  * BirmanSchwingerCore (birman_schwinger_core.py:523-540) and the quantised / mixed-precision wrappers
    (quantized_bk_core.py:168, mixed_precision_bk_core.py:131, complex_quantization.py:242): the batched get-diag slot
    with float64 operands and a complex128 0-dim z tensor, or with float16 operands.
Checked against the CPU oracle on the same values (1e-4), gradients against the oracle's reference-mode backward."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def bk():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import bkcore
    return bkcore


def _rel(got, ref):
    got, ref = np.asarray(got), np.asarray(ref)
    return float(np.max(np.abs(got - ref) / np.abs(ref)))


def test_non_hermitian_style_call(bk):
    from oracle import bk_oracle_torch as O
    torch.manual_seed(5)
    B, N = 4, 300
    V = torch.complex(-2.0 + 0.5 * torch.randn(B, N), -0.1 * torch.rand(B, N)).cuda().requires_grad_(True)
    h0_super = torch.nn.Parameter((1.0 + 0.1 * torch.randn(N - 1)).cuda())
    h0_sub = torch.nn.Parameter((1.0 + 0.1 * torch.randn(N - 1)).cuda())
    z = torch.nn.Parameter(torch.tensor(0.1 + 0.9j).cuda())
    he_diag = V.real                                           # a strided view: one float out of every two
    assert he_diag.stride(-1) == 2
    out = bk.BKCoreFunction.apply(he_diag, h0_super.unsqueeze(0).expand(B, -1), h0_sub.unsqueeze(0).expand(B, -1), z,
                                  False)
    feats, G = out
    gf = torch.randn(B, N, 2, device="cuda")
    (feats * gf).sum().backward()
    a = he_diag.detach().cpu()
    b = h0_super.detach().cpu().unsqueeze(0).expand(B, -1)
    c = h0_sub.detach().cpu().unsqueeze(0).expand(B, -1)
    f_ref, G_ref = O.forward(a, b, c, complex(z.detach().cpu()))
    assert _rel(G.detach().cpu().numpy(), G_ref.numpy()) <= TOL
    assert torch.equal(feats.detach(), torch.view_as_real(G.detach()))
    g_ref = O.backward_reference(G.detach().cpu(), gf.cpu(), None, bk.BKCoreFunction.GRAD_BLEND)
    got = V.grad.cpu()
    assert got.dtype == torch.complex64 and float(got.imag.abs().max()) == 0.0     # through .real: no imaginary part
    np.testing.assert_allclose(got.real.numpy(), g_ref.numpy(), rtol=TOL, atol=1e-5)
    assert h0_super.grad is None and h0_sub.grad is None and z.grad is None       # bk_core.py:384-385


def test_four_argument_call_with_expanded_buffers(bk):
    from oracle import bk_oracle_torch as O
    torch.manual_seed(6)
    B, N = 8, 777
    v = torch.clamp(0.5 * torch.randn(B, N), -3.0, 3.0).cuda().requires_grad_(True)
    h0_diag = torch.full((1, N), -2.0, device="cuda")
    h0_sub = torch.full((1, N - 1), 1.0, device="cuda")
    h0_super = torch.full((1, N - 1), 1.0, device="cuda")
    z = torch.tensor(1.0j, dtype=torch.complex64, device="cuda")
    he_diag = h0_diag.expand(B, -1) + v
    feats, G = bk.BKCoreFunction.apply(he_diag, h0_super.expand(B, -1), h0_sub.expand(B, -1), z)
    spec = torch.clamp(feats, -10.0, 10.0)                     # the callers clamp the features next
    spec.sum().backward()
    _, G_ref = O.forward(he_diag.detach().cpu(), h0_super.cpu().expand(B, -1), h0_sub.cpu().expand(B, -1), 1j)
    assert _rel(G.detach().cpu().numpy(), G_ref.numpy()) <= TOL
    g_ref = O.backward_reference(G.detach().cpu(), torch.ones(B, N, 2), None, bk.BKCoreFunction.GRAD_BLEND)
    np.testing.assert_allclose(v.grad.cpu().numpy(), g_ref.numpy(), rtol=TOL, atol=1e-5)


def test_get_diag_slot_float64_operands_complex128_z(bk):
    from oracle import bk_oracle_torch as O
    torch.manual_seed(7)
    B, N = 3, 2000
    a = (-2.0 + 0.5 * torch.randn(B, N)).double()
    b = (1.0 + 0.1 * torch.randn(B, N - 1)).double()
    c = (1.0 + 0.1 * torch.randn(B, N - 1)).double()
    z = torch.tensor(0.3 + 1.0j, dtype=torch.complex128, device="cuda")
    G = bk.vmapped_get_diag(a.cuda(), b.cuda(), c.cuda(), z)
    assert G.dtype == torch.complex64
    assert _rel(G.cpu().numpy(), O.get_diag(a, b, c, 0.3 + 1.0j).numpy()) <= TOL


def test_get_diag_slot_float16_operands(bk):
    from oracle import bk_oracle_torch as O
    torch.manual_seed(8)
    B, N = 5, 1500
    a = (-2.0 + 0.5 * torch.randn(B, N)).half()
    b = (1.0 + 0.1 * torch.randn(B, N - 1)).half()
    c = (1.0 + 0.1 * torch.randn(B, N - 1)).half()
    G = bk.vmapped_get_diag(a.cuda(), b.cuda(), c.cuda(), torch.tensor(1j, dtype=torch.complex64))   # z on the host
    assert _rel(G.cpu().numpy(), O.get_diag(a.float(), b.float(), c.float(), 1j).numpy()) <= TOL
