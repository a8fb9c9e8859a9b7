"""Every buffer the library writes -- G, features, the workspace, the state block, the boundary states, the gradients
and the analytic backward's workspace -- sits between two guard regions filled with a pattern; after the calls the
guards must be untouched.  Shapes are drawn so that every form of the forward (whole rows, two units a row, one
workgroup per chunk, scan forms, the double-precision chains), ragged lengths and both operand widths occur.  The calls
go through the C ABI directly, with buffers of exactly the sizes the library asks for."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
PAD = 4096
PATTERN = 0xA5


class Guarded:
    def __init__(self, nbytes, zero=False):
        self.n = int(nbytes)
        self.buf = torch.full((self.n + 2 * PAD,), PATTERN, dtype=torch.uint8, device="cuda")
        self.mid = self.buf[PAD:PAD + self.n]
        if zero:
            self.mid.zero_()

    def ptr(self):
        return self.mid.data_ptr()

    def intact(self):
        return bool((self.buf[:PAD] == PATTERN).all()) and bool((self.buf[PAD + self.n:] == PATTERN).all())

    def f32(self):
        return self.mid.view(torch.float32)


CASES = [  # B, N, dtype, flags (names of _lib constants), keep states
    (1, 1, "f32", (), True), (3, 2, "f32", (), True), (2, 17, "f32", (), True), (5, 4095, "f32", (), True),
    (4, 4096, "f32", (), True), (3, 4097, "f32", (), True), (7, 12289, "f32", (), True), (130, 8197, "f32", (), True),
    (700, 9000, "f32", (), True), (768, 12288, "f32", (), True), (1000, 16389, "f32", (), True),
    (1300, 16384, "f32", (), True), (1500, 9001, "f32", (), True), (16, 131073, "f32", (), True),
    (2, 300001, "f32", (), True), (200, 140000, "f32", (), False),
    (9, 20000, "f32", ("FLAG_RAW", "FLAG_PHI_EXACT"), True), (1000, 16384, "f32", ("FLAG_WHOLE_ROWS",), True),
    (300, 20481, "f32", ("FLAG_ROWS_UNCUT",), True), (133, 24591, "f32", ("FLAG_ROW_UNITS",), True),
    (20, 40000, "f32", ("FLAG_SCAN_PATH",), False), (6, 70001, "f32", ("FLAG_SCAN_PATH",), False),
    (600, 12300, "bf16", (), False), (1024, 8193, "bf16", (), False), (33, 50001, "bf16", ("FLAG_PHI_EXACT",), False),
]


@pytest.mark.parametrize("B,N,dtype,flag_names,keep", CASES)
def test_buffers_stay_inside_their_bounds(B, N, dtype, flag_names, keep):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from bkcore import _lib
    from oracle import bk_oracle_c as C
    L = _lib.load()
    flags = 0
    for n in flag_names:
        flags |= getattr(_lib, n)
    g = torch.Generator().manual_seed(B * 100003 + N)
    a = -2.0 + 0.5 * torch.randn(B, N, generator=g)
    b = 1.0 + 0.1 * torch.randn(B, max(N - 1, 1), generator=g)
    c = 1.0 + 0.1 * torch.randn(B, max(N - 1, 1), generator=g)
    td = torch.float32 if dtype == "f32" else torch.bfloat16
    dt = _lib.F32 if dtype == "f32" else _lib.BF16
    ad, bd, cd = a.to(td).cuda().contiguous(), b.to(td).cuda().contiguous(), c.to(td).cuda().contiguous()
    stream = torch.cuda.current_stream().cuda_stream
    nb = max(N - 1, 1)
    G = Guarded(B * N * 8)
    F = Guarded(B * N * 8)
    ws = Guarded(L.bkcore_workspace_bytes(B, N))
    state = Guarded(_lib.STATE_BYTES, zero=True)
    st_bytes = int(L.bkcore_boundary_state_bytes(B, N))
    states = Guarded(st_bytes)
    written = ctypes.c_int(0)
    z = (0.2, 1.0)
    for _ in range(2):   # twice: the second call finds the state block and the records as the first one left them
        if keep and dtype == "f32":
            rc = L.bkcore_forward_states(ad.data_ptr(), N, bd.data_ptr(), nb, cd.data_ptr(), nb, dt, None, z[0], z[1], B, N,
                                         G.ptr(), F.ptr(), 0.0, states.ptr(), st_bytes, ctypes.byref(written), ws.ptr(), ws.n,
                                         state.ptr(), flags, stream)
        else:
            rc = L.bkcore_forward_clamped(ad.data_ptr(), N, bd.data_ptr(), nb, cd.data_ptr(), nb, dt, None, z[0], z[1], B, N,
                                          G.ptr(), F.ptr(), 0.0, ws.ptr(), ws.n, state.ptr(), flags, stream)
        assert rc == 0, rc
    torch.cuda.synchronize()
    for name, buf in (("G", G), ("features", F), ("workspace", ws), ("state block", state), ("boundary states", states)):
        assert buf.intact(), f"{name}: guard region overwritten"
    # a too-small states buffer is refused, not overrun
    if keep and dtype == "f32" and st_bytes > 16:
        rc = L.bkcore_forward_states(ad.data_ptr(), N, bd.data_ptr(), nb, cd.data_ptr(), nb, dt, None, z[0], z[1], B, N,
                                     G.ptr(), F.ptr(), 0.0, states.ptr(), st_bytes - 16, ctypes.byref(written), ws.ptr(),
                                     ws.n, state.ptr(), flags, stream)
        assert rc == _lib.EWORKSPACE
    # the numbers: first and last row against the oracle (the operator's tolerance)
    rows = sorted({0, B - 1})
    Gh = G.f32().view(B, N, 2)[rows].cpu().numpy()
    raw = bool(flags & _lib.FLAG_RAW)
    exact = bool(flags & _lib.FLAG_PHI_EXACT)
    a64, b64, c64 = (x.to(td).double().numpy()[rows] for x in (a, b, c))
    if N == 1:
        b64, c64 = b64[:, :0], c64[:, :0]
    if raw:
        ref = C.get_diag(a64, b64, c64, complex(*z), phi_exact=exact)
    else:
        ref = C.forward_f32(a64, b64, c64, complex(*z), phi_exact=exact)[1]
    err = float(np.max(np.abs((Gh[..., 0] + 1j * Gh[..., 1]) - ref) / np.abs(ref)))
    assert err <= 1e-4, err
    assert torch.equal(G.f32(), F.f32())
    # backward kernels on the same guarded buffers
    gf = torch.randn(B, N, 2, generator=g).cuda()
    ga = Guarded(B * N * 4)
    rc = L.bkcore_backward_ref(G.ptr(), gf.data_ptr(), None, 0.5, B * N, ga.ptr(), stream)
    assert rc == 0
    torch.cuda.synchronize()
    assert ga.intact(), "grad_a of the reference-mode backward"
    if dtype == "f32" and not (flags & ~(_lib.FLAG_PHI_EXACT | _lib.FLAG_RAW)) and B * N <= (1 << 24):
        wsa = Guarded(L.bkcore_backward_analytic_workspace_bytes(B, N))
        gb, gc = Guarded(B * nb * 4), Guarded(B * nb * 4)
        use_states = bool(keep and written.value and not raw)
        # (the boundary states belong to the G of a forward with the same flags; RAW is a forward flag only)
        if raw:
            return
        rc = L.bkcore_backward_analytic_states(ad.data_ptr(), N, bd.data_ptr(), nb, cd.data_ptr(), nb, None, z[0], z[1], B, N,
                                               G.ptr(), states.ptr() if use_states else None, st_bytes if use_states else 0,
                                               gf.data_ptr(), None, ga.ptr(), gb.ptr() if N > 1 else None,
                                               gc.ptr() if N > 1 else None, wsa.ptr(), wsa.n, state.ptr(),
                                               _lib.FLAG_PHI_EXACT if exact else 0, stream)
        assert rc == 0, rc
        torch.cuda.synchronize()
        for name, buf in (("analytic workspace", wsa), ("grad_a", ga), ("grad_b", gb), ("grad_c", gc),
                          ("state block", state), ("G", G), ("boundary states", states)):
            assert buf.intact(), f"{name}: guard region overwritten by the analytic backward"
        assert bool(torch.isfinite(ga.f32()).all())


@pytest.mark.parametrize("outer,Lseq,inner,reverse", [(1, 1, 1, 0), (3, 1000, 1, 1), (7, 4097, 1, 0), (2, 70001, 1, 1),
                                                         (5, 333, 32, 0), (8, 2048, 32, 1), (3, 8191, 7, 0),
                                                         (130, 1025, 33, 1), (1, 50000, 3, 0)])
def test_cumsum_stays_inside_its_output(outer, Lseq, inner, reverse):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from bkcore import _lib
    L = _lib.load()
    g = torch.Generator().manual_seed(outer * 7 + Lseq)
    x = torch.randint(-3, 4, (outer, Lseq, inner), generator=g).float()
    xd = x.cuda().contiguous()
    y = Guarded(x.numel() * 4)
    rc = L.bkcore_cumsum(xd.data_ptr(), y.ptr(), outer, Lseq, inner, reverse, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    assert y.intact()
    ref = x.double().flip(1).cumsum(1).flip(1) if reverse else x.double().cumsum(1)
    assert torch.equal(y.f32().view(outer, Lseq, inner).cpu().double(), ref)     # integer-valued input: exact


@pytest.mark.parametrize("B,N,r", [(1, 5, 1), (3, 1000, 4), (16, 4097, 15), (64, 2049, 16), (9, 70001, 22), (130, 300, 7)])
def test_woodbury_stays_inside_its_buffers(B, N, r):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from bkcore import _lib
    L = _lib.load()
    g = torch.Generator().manual_seed(B + N + r)
    Gc = (torch.randn(B, N, 2, generator=g) * 0.3).cuda().contiguous()
    U = (torch.randn(N, r, generator=g) * 0.1).cuda().contiguous()
    V = (torch.randn(N, r, generator=g) * 0.1).cuda().contiguous()
    out = Guarded(B * N * 8)
    ws = Guarded(L.bkcore_woodbury_workspace_bytes(B, N, r))
    rc = L.bkcore_woodbury_correct(Gc.data_ptr(), U.data_ptr(), V.data_ptr(), B, N, r, out.ptr(), ws.ptr(), ws.n,
                                   torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
    torch.cuda.synchronize()
    assert out.intact() and ws.intact()
    # float64 restatement of the loop over the batch (birman_schwinger_core.py:567-604)
    Gz = (Gc[..., 0].double() + 1j * Gc[..., 1].double()).cpu()
    Uc, Vc = U.double().cpu().to(torch.complex128), V.double().cpu().to(torch.complex128)
    ref = torch.empty_like(Gz)
    for i in range(B):
        M = torch.eye(r, dtype=torch.complex128) + Vc.T @ (Gz[i][:, None] * Uc)
        W = torch.linalg.inv(M)
        ref[i] = Gz[i] - Gz[i] ** 2 * ((Uc @ W) * Vc).sum(-1)
    got = out.f32().view(B, N, 2).cpu().double()
    got = got[..., 0] + 1j * got[..., 1]
    assert float((got - ref).abs().max() / ref.abs().max()) <= 1e-4


@pytest.mark.parametrize("B,N,D", [(1, 1, 256), (3, 257, 256), (2, 1000, 512), (5, 77, 96), (8, 256, 1024)])
def test_layer_steps_stay_inside_their_buffers(B, N, D):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from bkcore import _lib
    L = _lib.load()
    stream = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(B * 31 + N)
    n = B * N
    v = torch.randn(B, N, generator=g).cuda()
    h0 = torch.full((N,), -2.0).cuda()
    he = Guarded(n * 2)
    assert L.bkcore_layer_prologue(v.data_ptr(), N, h0.data_ptr(), 3.0, B, N, he.ptr(), _lib.BF16, stream) == 0
    feats = (torch.randn(n, 2, generator=g) * 0.5).cuda()
    gfe = torch.randn(n, 2, generator=g).cuda()
    gv = Guarded(n * 4)
    assert L.bkcore_layer_backward_ref(feats.data_ptr(), gfe.data_ptr(), None, 0.5, 10.0, v.data_ptr(), 3.0, 1, n,
                                       gv.ptr(), stream) == 0
    w = torch.randn(D, 2, generator=g).cuda()
    bias = torch.randn(D, generator=g).cuda()
    scale = torch.randn(D, generator=g).cuda()
    res = torch.randn(n, D, generator=g).cuda()
    out = Guarded(n * D * 4)
    assert L.bkcore_layer_epilogue_clamped(feats.data_ptr(), 10.0, w.data_ptr(), bias.data_ptr(), scale.data_ptr(),
                                           res.data_ptr(), n, D, out.ptr(), stream) == 0
    go = torch.randn(n, D, generator=g).cuda()
    gfeat, gw, gb, gs = Guarded(n * 8), Guarded(D * 8), Guarded(D * 4), Guarded(D * 4)
    wsb = Guarded(L.bkcore_layer_epilogue_backward_workspace_bytes(n, D))
    assert L.bkcore_layer_epilogue_backward_clamped(go.data_ptr(), feats.data_ptr(), 10.0, w.data_ptr(), bias.data_ptr(),
                                                    scale.data_ptr(), n, D, gfeat.ptr(), gw.ptr(), gb.ptr(), gs.ptr(),
                                                    wsb.ptr(), wsb.n, stream) == 0
    torch.cuda.synchronize()
    for name, buf in (("he_diag", he), ("grad_v", gv), ("out", out), ("grad_features", gfeat), ("grad_weight", gw),
                      ("grad_bias", gb), ("grad_scale", gs), ("workspace", wsb)):
        assert buf.intact(), name
    ref = res + scale * (feats.clamp(-10, 10) @ w.T + bias)
    assert float((out.f32().view(n, D) - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
