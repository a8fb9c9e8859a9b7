"""Complex he_diag / h0_super / h0_sub for the raw resolvent diagonal (``vmapped_get_diag`` /
``get_tridiagonal_inverse_diagonal``), which the reference accepts by converting every operand to complex128
(src/models/bk_core.py:45-47), against outputs of the reference itself (tests/golden/bkcomplex_golden.npz,
oracle/gen_golden_complex.py).

CPU: the torch oracle against the fixture (float64 restatement: 1e-6).  GPU: ``bkcore_get_diag_complex`` -- the scan
kernel compiled with a complex k -- against the fixture and, at sizes beyond it, against the oracle: 1e-4 of each
element, as for real operands.  Rows on which nothing contracts hold stretches where |G| falls to 1e-8 and below by
cancellation; there float32 keeps 1e-4 of the row's typical magnitude rather than of the element (the same statement
as for real operands of that distribution), so the bound for elements below 1e-3 is absolute (1e-7)."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bkcomplex_golden.npz")
TOL = 1e-4


def _cases():
    z = np.load(GOLD, allow_pickle=False)
    return z, [str(n) for n in z["__names__"]]


def _err(got, ref):
    got, ref = np.asarray(got, np.complex128), np.asarray(ref, np.complex128)
    d = np.abs(got - ref)
    big = np.abs(ref) >= 1e-3
    rel = float((d[big] / np.abs(ref[big])).max()) if big.any() else 0.0
    small = float(d[~big].max()) if (~big).any() else 0.0
    return rel, small


def test_fixture_contents():
    z, names = _cases()
    assert len(names) >= 8 and bool(z["operator_refuses_complex"])
    kinds = set()
    for n in names:
        a, b, c, G = (z[f"{n}/{k}"] for k in ("a", "b", "c", "G"))
        assert G.dtype == np.complex64 and G.shape == a.shape and b.shape == c.shape == (a.shape[0], a.shape[1] - 1)
        kinds.add((np.iscomplexobj(a), np.iscomplexobj(b)))
        assert np.isfinite(G.view(np.float32)).all() and np.abs(G).max() <= 50.0 + 1e-4
    assert kinds == {(True, False), (False, True), (True, True)}      # complex diagonal only, off-diagonals only, all


def test_oracle_reproduces_reference_on_complex_operands():
    from oracle import bk_oracle_torch as O
    z, names = _cases()
    for n in names:
        a, b, c = (torch.from_numpy(z[f"{n}/{k}"]) for k in ("a", "b", "c"))
        G = O.get_diag(a, b, c, complex(*z[f"{n}/z"]))
        rel, small = _err(G.numpy(), z[f"{n}/G"])
        assert rel <= 1e-6 and small <= 1e-9, (n, rel, small)


def test_operator_refuses_complex_operands_as_the_reference_does():
    """BKCoreFunction clamps its inputs first (bk_core.py:256-258) and torch has no clamp for complex tensors: the
    reference raises, and so does the operator here -- before anything touches a device."""
    import bkcore
    a = torch.zeros(1, 4, dtype=torch.complex64)
    b = torch.ones(1, 3)
    with pytest.raises(TypeError, match="complex"):
        bkcore.BKCoreFunction.apply(a, b, b, 1j)
    with pytest.raises(TypeError, match="complex"):
        bkcore.bk_forward(a, b, b, 1j)


@pytest.fixture(scope="module")
def bk():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import bkcore
    return bkcore


@pytest.mark.gpu
def test_golden_complex_get_diag(bk):
    z, names = _cases()
    for n in names:
        a, b, c = (torch.from_numpy(z[f"{n}/{k}"]).cuda() for k in ("a", "b", "c"))
        zz = complex(*z[f"{n}/z"])
        G = bk.vmapped_get_diag(a, b, c, zz)
        assert G.dtype == torch.complex64 and G.shape == a.shape
        rel, small = _err(G.cpu().numpy(), z[f"{n}/G"])
        assert rel <= TOL and small <= 1e-7, (n, rel, small)
        g0 = bk.get_tridiagonal_inverse_diagonal(a[0], b[0], c[0], torch.tensor(zz, dtype=torch.complex64, device="cuda"))
        assert torch.equal(g0, G[0])                     # one row, z as a device tensor: the same launch


def _seeded(B, N, seed, im=0.3):
    g = torch.Generator().manual_seed(seed)
    cp = lambda shape, m, s, t: torch.complex(m + s * torch.randn(shape, generator=g), t * torch.randn(shape, generator=g))
    return cp((B, N), -2.0, 0.5, im), cp((B, N - 1), 1.0, 0.1, 0.2), cp((B, N - 1), 1.0, 0.1, 0.2)


@pytest.mark.gpu
@pytest.mark.parametrize("B,N", [(3, 2047), (2, 2048), (2, 2049), (1, 6150), (5, 8192), (2, 20000)])
@pytest.mark.parametrize("phi", ["reference", "exact"])
def test_seeded_complex_vs_oracle(bk, B, N, phi):
    """Lengths around the complex kernel's 2048-element chunks, both phi conventions."""
    from oracle import bk_oracle_torch as O
    a, b, c = _seeded(B, N, 7000 + N)
    ref = O.get_diag(a, b, c, 0.2 + 0.9j, phi_k_index=phi).numpy()
    bk.BKCoreFunction.PHI_K_INDEX = phi
    try:
        G = bk.vmapped_get_diag(a.cuda(), b.cuda(), c.cuda(), 0.2 + 0.9j)
    finally:
        bk.BKCoreFunction.PHI_K_INDEX = "reference"
    rel, small = _err(G.cpu().numpy(), ref)
    assert rel <= TOL and small <= 1e-7, (rel, small)


@pytest.mark.gpu
def test_complex_with_zero_imaginary_parts_equals_real_path(bk):
    """Complex operands whose imaginary parts are zero give the real-operand result (to the tolerance: another kernel
    form computes it); mixed real / complex operands are converted like the reference converts them."""
    g = torch.Generator().manual_seed(11)
    a = (-2.0 + 0.5 * torch.randn(4, 5000, generator=g)).cuda()
    b = (1.0 + 0.1 * torch.randn(4, 4999, generator=g)).cuda()
    c = (1.0 + 0.1 * torch.randn(4, 4999, generator=g)).cuda()
    Gr = bk.vmapped_get_diag(a, b, c, 1j)
    Gc = bk.vmapped_get_diag(a.to(torch.complex64), b, c.to(torch.complex128), 1j)
    rel, small = _err(Gc.cpu().numpy(), Gr.cpu().numpy())
    assert rel <= TOL and small <= 1e-7


@pytest.mark.gpu
def test_complex_broadcast_rows_strides_and_conjugated_views(bk):
    from oracle import bk_oracle_torch as O
    a, b, c = _seeded(6, 3000, 21)
    b1, c1 = b[:1], c[:1]
    ref = O.get_diag(a, b1.expand(6, -1), c1.expand(6, -1), 0.5 + 0.8j).numpy()
    ad = a.cuda()
    G = bk.vmapped_get_diag(ad, b1.cuda().expand(6, -1), c1.cuda().expand(6, -1), 0.5 + 0.8j)      # leading dimension 0
    rel, small = _err(G.cpu().numpy(), ref)
    assert rel <= TOL and small <= 1e-7
    wide = torch.zeros(6, 3100, dtype=torch.complex64, device="cuda")
    wide[:, :3000] = ad
    G2 = bk.vmapped_get_diag(wide[:, :3000], b1.cuda().expand(6, -1), c1.cuda().expand(6, -1), 0.5 + 0.8j)
    assert torch.equal(G2, G)                                                                      # row stride 3100
    G3 = bk.vmapped_get_diag(ad.conj().conj(), b1.cuda().expand(6, -1), c1.cuda().expand(6, -1), 0.5 + 0.8j)
    assert torch.equal(G3, G)
    refc = O.get_diag(a.conj().resolve_conj(), b1.expand(6, -1), c1.expand(6, -1), 0.5 + 0.8j).numpy()
    G4 = bk.vmapped_get_diag(ad.conj(), b1.cuda().expand(6, -1), c1.cuda().expand(6, -1), 0.5 + 0.8j)   # lazy conjugate
    rel, small = _err(G4.cpu().numpy(), refc)
    assert rel <= TOL and small <= 1e-7


@pytest.mark.gpu
def test_complex_entry_point_arguments(bk):
    """Through the C ABI directly: empty problems, a workspace that is too small, flags it does not take, and memory
    past G and on either side of the workspace left alone."""
    from bkcore import _lib
    L = _lib.load()
    a, b, c = (x.cuda() for x in _seeded(3, 2500, 5))
    need = int(L.bkcore_get_diag_complex_workspace_bytes(3, 2500))
    assert need > 0 and L.bkcore_get_diag_complex_workspace_bytes(0, 5) == 0
    guard = 4096
    G = torch.full((3 * 2500 + guard,), 7.0 + 7.0j, dtype=torch.complex64, device="cuda")
    ws = torch.full((need + 2 * guard,), 0x5A, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    call = lambda B, N, wsb, flags: L.bkcore_get_diag_complex(
        a.data_ptr(), 2500, b.data_ptr(), 2499, c.data_ptr(), 2499, None, 0.0, 1.0, B, N, G.data_ptr(),
        ws.data_ptr() + guard, wsb, flags, stream)
    assert call(0, 2500, need, 0) == 0 and call(3, 0, need, 0) == 0
    assert call(3, 2500, need - 1, 0) == _lib.EWORKSPACE
    assert call(3, 2500, need, _lib.FLAG_RAW) == _lib.EINVAL and call(-1, 2500, need, 0) == _lib.EINVAL
    torch.cuda.synchronize()
    assert bool((G == 7.0 + 7.0j).all())                       # nothing above launched anything
    assert call(3, 2500, need, 0) == 0
    torch.cuda.synchronize()
    assert bool((G[3 * 2500:] == 7.0 + 7.0j).all())
    assert bool((ws[:guard] == 0x5A).all()) and bool((ws[guard + need:] == 0x5A).all())
    from oracle import bk_oracle_torch as O
    ref = O.get_diag(a.cpu(), b.cpu(), c.cpu(), 1j).numpy()
    rel, small = _err(G[:3 * 2500].reshape(3, 2500).cpu().numpy(), ref)
    assert rel <= TOL and small <= 1e-7
