"""GPU parity tests: HIP kernels (through the C ABI / the operator wrapper) against the CPU oracle and
against the golden fixtures recorded from the reference.

Tolerance (stated by the task's north star): max relative error <= 1e-4 on the complex64 G and on the
gradients, measured elementwise as |x - ref| / |ref|.
"""
import numpy as np
import pytest
import torch

from conftest import max_rel_err
from test_oracle_golden import ALL

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope="module")
def bk():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import bkcore
    return bkcore


@pytest.fixture(scope="module")
def coracle():
    from oracle import bk_oracle_c as C
    C.lib()
    return C


def _z(cs):
    return complex(cs["z"][0], cs["z"][1])


def _dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def _lap(B, N, seed, varying=False):
    g = torch.Generator().manual_seed(seed)
    a = -2.0 + 0.5 * torch.randn(B, N, generator=g)
    if varying:
        b = 1.0 + 0.1 * torch.randn(B, N - 1, generator=g)
        c = 1.0 + 0.1 * torch.randn(B, N - 1, generator=g)
    else:
        b = torch.ones(B, N - 1)
        c = torch.ones(B, N - 1)
    return a, b, c


@pytest.mark.parametrize("name", ALL)
def test_golden_forward(bk, golden, name):
    cs = golden[name]
    a, b, c = _dev(cs["a"]), _dev(cs["b"]), _dev(cs["c"])
    z = torch.tensor(_z(cs), dtype=torch.complex64, device="cuda")
    feats, G = bk.BKCoreFunction.apply(a, b, c, z)
    assert feats.shape == cs["features"].shape and feats.dtype == torch.float32
    assert G.shape == cs["G"].shape and G.dtype == torch.complex64
    Gh = G.cpu().numpy()
    ref = cs["G"]
    zero = ref == 0
    assert np.array_equal(Gh == 0, zero)
    assert max_rel_err(Gh[~zero], ref[~zero]) <= TOL
    assert np.array_equal(feats.cpu().numpy(), np.stack([Gh.real, Gh.imag], -1))


@pytest.mark.parametrize("name", [n for n in ALL if n not in ("poisoned", "nan_offdiag_only", "clamped_inputs")])
def test_golden_raw_get_diag(bk, golden, name):
    cs = golden[name]
    G = bk.vmapped_get_diag(_dev(cs["a"]), _dev(cs["b"]), _dev(cs["c"]), _z(cs)).cpu().numpy()
    assert max_rel_err(G, cs["G_raw"]) <= TOL


@pytest.mark.parametrize("name", ALL)
@pytest.mark.parametrize("alpha", [0.0, 0.5, 1.0])
@pytest.mark.parametrize("with_gG", [False, True])
def test_golden_reference_backward(bk, golden, name, alpha, with_gG):
    # the backward kernel alone, fed the reference's own saved G: isolates the elementwise formula
    cs = golden[name]
    got = bk.bk_backward_reference(_dev(cs["G"]), _dev(cs["gfeat"]), _dev(cs["gG"]) if with_gG else None, alpha)
    ref = cs[f"grad_ref_a{alpha}_gG{int(with_gG)}"]
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-4, atol=1e-5)


def test_autograd_function_end_to_end(bk, golden):
    # forward + backward through torch autograd, as the reference's test does (tests/test_bk_core.py:130-158)
    cs = golden["lap_n64_b8"]
    a = _dev(cs["a"]).requires_grad_(True)
    z = torch.tensor(1j, dtype=torch.complex64, device="cuda")
    gf = _dev(cs["gfeat"])
    for alpha in (0.0, 0.5, 1.0):
        bk.BKCoreFunction.GRAD_BLEND = alpha
        a.grad = None
        feats, G = bk.BKCoreFunction.apply(a, _dev(cs["b"]), _dev(cs["c"]), z)
        (feats * gf).sum().backward()
        ref = cs[f"grad_ref_a{alpha}_gG0"]
        got = a.grad.cpu().numpy()
        # the reference gradient has terms ~1/G^2, so a 1e-5 relative difference in G moves it by ~2e-5
        err = float(np.max(np.abs(got - ref) / (np.abs(ref) + 1e-3 * np.abs(ref).max())))
        print(f"alpha={alpha}: gradient error {err:.2e}")
        assert err <= TOL
    bk.BKCoreFunction.GRAD_BLEND = 0.5
    assert a.grad.dtype == torch.float32 and a.grad.shape == a.shape


@pytest.mark.parametrize("B,N", [(8, 1024), (3, 2047), (3, 2048), (3, 2049), (2, 4097), (5, 300), (4, 8192), (2, 32768)])
@pytest.mark.parametrize("varying", [False, True])
def test_seeded_vs_c_oracle(bk, coracle, B, N, varying):
    a, b, c = _lap(B, N, 77 + N + B, varying)
    z = 1j
    Gref = coracle.get_diag(a.numpy(), b.numpy(), c.numpy(), z)
    G = bk.vmapped_get_diag(a.cuda(), b.cuda(), c.cuda(), z).cpu().numpy()
    assert max_rel_err(G, Gref) <= TOL
    if varying:
        bk.BKCoreFunction.PHI_K_INDEX = "exact"
        try:
            Ge = bk.vmapped_get_diag(a.cuda(), b.cuda(), c.cuda(), z).cpu().numpy()
        finally:
            bk.BKCoreFunction.PHI_K_INDEX = "reference"
        Geref = coracle.get_diag(a.numpy(), b.numpy(), c.numpy(), z, phi_exact=True)
        assert max_rel_err(Ge, Geref) <= TOL


@pytest.mark.parametrize("N", [1, 2, 3, 7, 8, 9, 63, 64, 65, 255, 256, 257, 511, 513])
def test_small_and_ragged_lengths(bk, coracle, N):
    g = torch.Generator().manual_seed(N)
    B = 3
    a = -2.0 + 0.5 * torch.randn(B, N, generator=g)
    b = 1.0 + 0.1 * torch.randn(B, max(N - 1, 0), generator=g)
    c = 1.0 + 0.1 * torch.randn(B, max(N - 1, 0), generator=g)
    z = 0.3 + 0.9j
    feats, G = bk.BKCoreFunction.apply(a.cuda(), b.cuda(), c.cuda(), z)
    F, Gref = coracle.forward_f32(a.numpy(), b.numpy(), c.numpy(), z)
    assert max_rel_err(G.cpu().numpy(), Gref) <= TOL


@pytest.mark.parametrize("case", ["in_range_large_sums", "diag_outside", "offdiag_outside", "infinite_diag", "nan_diag"])
def test_staging_shortcut_boundaries(bk, coracle, case):
    """The staging step leaves the clamps out for a wave whose sums of squares prove every value inside them.  Values
    inside the clamps whose sums are large, single values outside them, an infinity and a NaN must all still give the
    operator's results (bk_core.py:249-264)."""
    g = torch.Generator().manual_seed(len(case))
    B, N = 3, 8192 + 100
    a = -8.0 + 0.5 * torch.randn(B, N, generator=g)
    b = 1.0 + 0.1 * torch.randn(B, N - 1, generator=g)
    c = 1.0 + 0.1 * torch.randn(B, N - 1, generator=g)
    if case == "in_range_large_sums":
        a -= 22.0
        b *= 3.0
        c *= -3.0
    elif case == "diag_outside":
        a[1, 5000] = 250.0
        a[2, 70] = -1.0e6
        a[2, N - 1] = 101.0
    elif case == "offdiag_outside":
        b[0, 4100] = 25.0
        c[1, 8000] = -40.0
        b[2, 0] = 10.5
    elif case == "infinite_diag":
        a[0, 3000] = float("inf")
        a[1, 3] = float("-inf")
    elif case == "nan_diag":
        a[0, 6000] = float("nan")
    z = 0.3 + 0.9j
    feats, G = bk.BKCoreFunction.apply(a.cuda(), b.cuda(), c.cuda(), z)
    F, Gref = coracle.forward_f32(a.numpy(), b.numpy(), c.numpy(), z)
    assert np.isfinite(Gref).all()
    assert max_rel_err(G.cpu().numpy(), Gref) <= TOL
    # the same rows in one batch with rows that take the shortcut: bit for bit the same results
    a2 = torch.cat([a, -2.0 + 0.5 * torch.randn(2, N, generator=g)])
    b2 = torch.cat([b, torch.ones(2, N - 1)])
    c2 = torch.cat([c, torch.ones(2, N - 1)])
    G2 = bk.BKCoreFunction.apply(a2.cuda(), b2.cuda(), c2.cuda(), z)[1]
    assert torch.equal(G2[:B], G)


def test_empty_inputs(bk):
    z = 1j
    feats, G = bk.BKCoreFunction.apply(torch.zeros(4, 0, device="cuda"), torch.zeros(4, 0, device="cuda"),
                                       torch.zeros(4, 0, device="cuda"), z)
    assert G.shape == (4, 0) and feats.shape == (4, 0, 2)
    feats, G = bk.BKCoreFunction.apply(torch.zeros(0, 16, device="cuda"), torch.zeros(0, 15, device="cuda"),
                                       torch.zeros(0, 15, device="cuda"), z)
    assert G.shape == (0, 16) and feats.shape == (0, 16, 2)


def test_broadcast_rows_and_dtypes(bk, coracle):
    # the layer passes expand()ed (stride-0) off-diagonals and bf16 inputs (resnet_bk.py:155-170)
    B, N = 6, 700
    g = torch.Generator().manual_seed(5)
    a = (1.0 + 5.0 * torch.tanh(torch.randn(B, N, generator=g) / 5.0))
    b1 = 0.5 + 0.1 * torch.randn(1, N - 1, generator=g)
    c1 = 0.7 + 0.1 * torch.randn(1, N - 1, generator=g)
    z = torch.tensor(1.3133j, dtype=torch.complex64)
    for dt in (torch.bfloat16, torch.float16, torch.float32, torch.float64):
        ad, bd, cd = a.to(dt), b1.to(dt), c1.to(dt)
        _, Gref = coracle.forward_f32(ad.float().numpy(), bd.float().expand(B, -1).numpy(),
                                      cd.float().expand(B, -1).numpy(), complex(z))
        feats, G = bk.BKCoreFunction.apply(ad.cuda(), bd.cuda().expand(B, -1), cd.cuda().expand(B, -1), z.cuda())
        assert max_rel_err(G.cpu().numpy(), Gref) <= TOL, dt


def test_z_forms_agree(bk):
    a, b, c = _lap(4, 500, 3)
    a, b, c = a.cuda(), b.cuda(), c.cuda()
    z = 0.25 + 0.8j
    G0 = bk.vmapped_get_diag(a, b, c, z)
    G1 = bk.vmapped_get_diag(a, b, c, torch.tensor(z, dtype=torch.complex64))
    G2 = bk.vmapped_get_diag(a, b, c, torch.tensor(z, dtype=torch.complex64, device="cuda"))
    G3 = bk.vmapped_get_diag(a, b, c, torch.tensor(z, dtype=torch.complex128, device="cuda"))
    assert torch.equal(G0, G1) and torch.equal(G0, G2) and torch.equal(G0, G3)
    row = bk.get_tridiagonal_inverse_diagonal(a[1], b[1], c[1], z)
    assert torch.equal(row, G0[1])


def test_headline_size_properties(bk, coracle):
    """B=1024, N=32768 (the headline configuration): parity on a seeded row subset against the oracle,
    row independence, features == G, bitwise determinism."""
    B, N = 1024, 32768
    g = torch.Generator(device="cuda").manual_seed(1234)
    a = -2.0 + 0.5 * torch.randn(B, N, generator=g, device="cuda")
    b = torch.ones(B, N - 1, device="cuda")
    c = torch.ones(B, N - 1, device="cuda")
    z = torch.tensor(1j, dtype=torch.complex64, device="cuda")
    feats, G = bk.BKCoreFunction.apply(a, b, c, z)
    feats2, G2 = bk.BKCoreFunction.apply(a, b, c, z)
    assert torch.equal(G, G2) and torch.equal(feats, feats2)
    assert torch.equal(torch.view_as_real(G), feats)
    assert torch.isfinite(feats).all()
    rows = [0, 1, 511, 777, 1023]
    Gref = coracle.get_diag(a[rows].cpu().numpy(), np.ones((len(rows), N - 1)), np.ones((len(rows), N - 1)), 1j)
    assert max_rel_err(G[rows].cpu().numpy(), Gref) <= TOL
    # a row computed inside another batch equals the same row computed inside this one (rows are independent systems):
    # bitwise where the batch takes the same form (1000 rows: two units a row, like 1024), within tolerance where the
    # running product is put together in another order (300 rows: one workgroup per chunk; one row alone likewise)
    _, Gsub = bk.BKCoreFunction.apply(a[24:1024], b[24:1024], c[24:1024], z)
    assert torch.equal(Gsub[753], G[777])
    _, Gsub = bk.BKCoreFunction.apply(a[600:900], b[600:900], c[600:900], z)
    assert max_rel_err(Gsub[177].cpu().numpy(), G[777].cpu().numpy()) <= 5e-5
    _, Gone = bk.BKCoreFunction.apply(a[777:778], b[777:778], c[777:778], z)
    assert max_rel_err(Gone[0].cpu().numpy(), G[777].cpu().numpy()) <= 5e-5   # two evaluation orders, each ~1.5e-5 from the oracle
    # backward at the headline size: elementwise formula against the oracle's on the same rows
    gf = torch.randn(B, N, 2, generator=g, device="cuda")
    grad = bk.bk_backward_reference(G, gf, None, 0.5)
    ref = coracle.backward_ref(G[rows].cpu().numpy(), gf[rows].cpu().numpy(), None, 0.5)
    np.testing.assert_allclose(grad[rows].cpu().numpy(), ref, rtol=1e-4, atol=1e-5)


def test_long_sequence(bk, coracle):
    """N = 2^20 (long-context configuration, B reduced to keep the oracle to seconds)."""
    from bkcore import bk_core
    B, N = 2, 1 << 20
    a, b, c = _lap(B, N, 99)
    Gref = coracle.get_diag(a.numpy(), b.numpy(), c.numpy(), 1j)
    bk_core._DEBUG["keep_ws"] = True
    try:
        G = bk.vmapped_get_diag(a.cuda(), b.cuda(), c.cuda(), 1j).cpu().numpy()
        redone = bk_core.rows_recomputed()
    finally:
        bk_core._DEBUG["keep_ws"] = False
        bk_core._DEBUG["last_ws"] = None
    assert max_rel_err(G, Gref) <= TOL
    assert redone == 0


@pytest.mark.parametrize("B,N,dtype,conv", [(16, 65536 + 137, torch.float32, "reference"),
                                            (3, 40000, torch.float32, "exact"),
                                            (5, 8 * 4096, torch.bfloat16, "reference"),
                                            (1, 300001, torch.float64, "reference")])
@pytest.mark.parametrize("scan_path", [False, True])
def test_few_long_rows_use_several_workgroups_per_row(bk, coracle, B, N, dtype, conv, scan_path):
    """B <= 128 and N >= 8 chunks: rows are cut into segments (one workgroup each), in ratio form by default and in
    scan form with BKCORE_FLAG_SCAN_PATH.  Parity against the oracle and no row falling back to the one-workgroup
    kernel on well-conditioned input."""
    from bkcore import bk_core
    a, b, c = _lap(B, N, 17 + B, varying=True)
    a, b, c = a.to(dtype), b.to(dtype), c.to(dtype)
    z = 0.3 + 0.9j
    bk_core._DEBUG["keep_ws"] = True
    bk_core._DEBUG["scan_path"] = scan_path
    bk.BKCoreFunction.PHI_K_INDEX = conv
    try:
        feats, G = bk.BKCoreFunction.apply(a.cuda(), b.cuda(), c.cuda(), z)
        torch.cuda.synchronize()
        redone = bk_core.rows_recomputed()
    finally:
        bk.BKCoreFunction.PHI_K_INDEX = "reference"
        bk_core._DEBUG["keep_ws"] = False
        bk_core._DEBUG["scan_path"] = False
        bk_core._DEBUG["last_ws"] = None
    Gref = coracle.get_diag(a.double().numpy(), b.double().numpy(), c.double().numpy(), z, phi_exact=(conv == "exact"))
    assert max_rel_err(G.cpu().numpy(), Gref) <= TOL
    assert torch.equal(torch.view_as_real(G), feats)
    assert redone == 0


def test_few_long_rows_fall_back_when_ill_conditioned(bk):
    """The reference benchmark's randn distribution (scripts/benchmark_bk_triton.py:66-70) is not contractive: the
    segment checks fail and the rows are recomputed by the one-workgroup kernel -- bitwise the result the same rows
    get inside a large batch (which never takes the several-workgroups path)."""
    from bkcore import bk_core
    g = torch.Generator().manual_seed(3)
    B, N = 300, 8 * 4096 + 5
    a = torch.randn(B, N, generator=g).cuda()
    b = torch.randn(B, N - 1, generator=g).cuda()
    c = torch.randn(B, N - 1, generator=g).cuda()
    z = 0.1 + 0.1j
    Gbig = bk.vmapped_get_diag(a, b, c, z)
    bk_core._DEBUG["keep_ws"] = True
    try:
        Gfew = bk.vmapped_get_diag(a[:3], b[:3], c[:3], z)
        torch.cuda.synchronize()
        redone = bk_core.rows_recomputed()
    finally:
        bk_core._DEBUG["keep_ws"] = False
        bk_core._DEBUG["last_ws"] = None
    assert redone == 3
    assert torch.equal(Gfew, Gbig[:3])


@pytest.mark.parametrize("N", [1, 5, 16, 17, 4095, 4096, 4097, 4112, 4113, 8192, 8193, 8208, 8209, 12288, 20000])
def test_ratio_form_and_scan_first_launch_agree(bk, coracle, N):
    """The default first launch (single sweep in ratio form) and the scan-based one (BKCORE_FLAG_SCAN_PATH) are two
    evaluation orders of the same recurrences: both within tolerance of the oracle, neither recomputing a row on
    well-conditioned input.  Lengths straddle chunk (4096) and halo (16) boundaries."""
    from bkcore import bk_core
    B = 130                                                      # > 128 rows: one workgroup per row
    a, b, c = _lap(B, N, 1000 + N, varying=True)
    z = 0.3 + 0.9j
    out = {}
    bk_core._DEBUG["keep_ws"] = True
    try:
        for scan in (False, True):
            bk_core._DEBUG["scan_path"] = scan
            G = bk.vmapped_get_diag(a.cuda(), b.cuda(), c.cuda(), z)
            torch.cuda.synchronize()
            assert bk_core.rows_recomputed() == 0
            out[scan] = G.cpu().numpy()
    finally:
        bk_core._DEBUG["scan_path"] = False
        bk_core._DEBUG["keep_ws"] = False
        bk_core._DEBUG["last_ws"] = None
    rows = [0, 64, B - 1]
    Gref = coracle.get_diag(a[rows].numpy(), b[rows].numpy(), c[rows].numpy(), z)
    assert max_rel_err(out[False][rows], Gref) <= TOL
    assert max_rel_err(out[True][rows], Gref) <= TOL
    assert max_rel_err(out[False], out[True]) <= TOL


def test_ratio_form_falls_back_when_not_contracting(bk):
    """randn a, b, c with z close to the real axis: the ratio maps do not contract, every row fails its checks and is
    recomputed by the two-scan kernel -- the same kernel, hence the same bits, whichever first launch ran."""
    from bkcore import bk_core
    g = torch.Generator().manual_seed(8)
    B, N = 160, 6000
    a = torch.randn(B, N, generator=g).cuda()
    b = torch.randn(B, N - 1, generator=g).cuda()
    c = torch.randn(B, N - 1, generator=g).cuda()
    z = 0.1 + 0.1j
    bk_core._DEBUG["keep_ws"] = True
    try:
        G0 = bk.vmapped_get_diag(a, b, c, z)
        torch.cuda.synchronize()
        redone = bk_core.rows_recomputed()
        bk_core._DEBUG["scan_path"] = True
        G1 = bk.vmapped_get_diag(a, b, c, z)
        torch.cuda.synchronize()
        redone_scan = bk_core.rows_recomputed()
    finally:
        bk_core._DEBUG["scan_path"] = False
        bk_core._DEBUG["keep_ws"] = False
        bk_core._DEBUG["last_ws"] = None
    assert redone == B and redone_scan == B
    assert torch.equal(G0, G1)


def test_too_long_is_rejected(bk):
    from bkcore import _lib
    nmax = int(_lib.load().bkcore_max_seq_len())
    a = torch.zeros(1, nmax + 1, device="cuda")
    with pytest.raises(RuntimeError, match="sequence length"):
        bk.vmapped_get_diag(a, a[:, :-1], a[:, :-1], 1j)


def test_cpu_tensors_fail_loudly(bk):
    a = torch.zeros(1, 8)
    with pytest.raises(RuntimeError, match="ROCm device"):
        bk.vmapped_get_diag(a, a[:, :-1], a[:, :-1], 1j)


def test_batch_rows_of_off_diagonals_are_validated(bk):
    """h0_super / h0_sub with a row count that is neither the batch nor 1 would be read out of bounds."""
    a = torch.zeros(4, 16, device="cuda")
    good = torch.ones(4, 15, device="cuda")
    for bad in (torch.ones(2, 15, device="cuda"), torch.ones(3, 15, device="cuda")):
        with pytest.raises(ValueError, match="rows for a batch"):
            bk.BKCoreFunction.apply(a, bad, good, 1j)
        with pytest.raises(ValueError, match="rows for a batch"):
            bk.bk_backward_analytic(a, good, bad, 1j, torch.zeros(4, 16, 2, device="cuda"), None)
    bk.BKCoreFunction.apply(a - 2.0, good[:1], good[:1].expand(4, -1), 1j)     # one row and expand() are fine


def test_config3_exact_shape(bk, coracle):
    """BASELINE configs[2] at its own shape, B = 16 rows of N = 2^20 (rows cut into segments, double-precision chains):
    two rows against the oracle, no row recomputed, bitwise determinism."""
    from bkcore import bk_core
    B, N = 16, 1 << 20
    g = torch.Generator(device="cuda").manual_seed(2020)
    a = -2.0 + 0.5 * torch.randn(B, N, generator=g, device="cuda")
    b = torch.ones(B, N - 1, device="cuda")
    c = torch.ones(B, N - 1, device="cuda")
    z = torch.tensor(1j, dtype=torch.complex64, device="cuda")
    bk_core._DEBUG["keep_ws"] = True
    try:
        feats, G = bk.BKCoreFunction.apply(a, b, c, z)
        torch.cuda.synchronize()
        assert bk_core.rows_recomputed() == 0
    finally:
        bk_core._DEBUG["keep_ws"] = False
        bk_core._DEBUG["last_ws"] = None
    rows = [0, 15]
    ref = coracle.forward_f32(a[rows].cpu().numpy(), b[rows].cpu().numpy(), c[rows].cpu().numpy(), 1j)[1]
    assert max_rel_err(G[rows].cpu().numpy(), ref) <= TOL
    assert torch.equal(torch.view_as_real(G), feats)
    _, G2 = bk.BKCoreFunction.apply(a, b, c, z)
    assert torch.equal(torch.view_as_real(G2), torch.view_as_real(G))


def test_config4_rows_on_one_gpu(bk, coracle):
    """BASELINE configs[3]'s whole batch, B = 8192 rows of N = 32768, on one GPU: a row subset against the oracle,
    features == G, and the backward on the same rows."""
    B, N = 8192, 32768
    g = torch.Generator(device="cuda").manual_seed(4040)
    a = -2.0 + 0.5 * torch.randn(B, N, generator=g, device="cuda")
    bc = torch.ones(1, N - 1, device="cuda").expand(B, -1)      # stride-0 batch dimension, as the layer's buffers
    z = torch.tensor(1j, dtype=torch.complex64, device="cuda")
    feats, G = bk.BKCoreFunction.apply(a, bc, bc, z)
    rows = [0, 4095, 8191]
    ones = np.ones((len(rows), N - 1), np.float32)
    ref = coracle.forward_f32(a[rows].cpu().numpy(), ones, ones, 1j)[1]
    assert max_rel_err(G[rows].cpu().numpy(), ref) <= TOL
    assert torch.equal(torch.view_as_real(G), feats)
    gf = torch.randn(len(rows), N, 2, generator=g, device="cuda")
    grad = bk.bk_backward_reference(G[rows], gf, None, 0.5)
    gref = coracle.backward_ref(G[rows].cpu().numpy(), gf.cpu().numpy(), None, 0.5)
    np.testing.assert_allclose(grad.cpu().numpy(), gref, rtol=1e-4, atol=1e-5)
    del feats, G, a
    torch.cuda.empty_cache()
