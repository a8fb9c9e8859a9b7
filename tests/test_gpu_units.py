"""Rows cut into units that hand the running product on inside the first launch (BKCORE_FLAG_ROW_UNITS; taken by the
library itself for float32 / float64 operands in a band of row counts a little above what the GPU holds at once).  Every chunk is computed by the same
code from the same numbers whichever workgroup runs it, so the results must equal those of whole rows bit for bit --
on every path of the operator: ragged lengths, 16-bit and float64 operands, the reference's NaN rules
(bk_core.py:256-264), rows whose checks fail and are recomputed, graph replay."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CH = 4096  # elements per chunk of the ratio-form kernel
UNCUT = 256  # BKCORE_FLAG_ROWS_UNCUT


@pytest.fixture()
def bk():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import bkcore
    from bkcore import bk_core
    bk_core._DEBUG["keep_ws"] = True
    yield bkcore
    bk_core._DEBUG["keep_ws"] = False
    bk_core._DEBUG["row_units"] = 0


def _both(bk, units_flag, *args, **kw):
    from bkcore import bk_core, _lib
    bk_core._DEBUG["row_units"] = UNCUT           # one workgroup per row whatever the shape
    f0, G0 = bk.bk_forward(*args, **kw)
    r0 = bk_core.rows_recomputed()
    bk_core._DEBUG["row_units"] = units_flag
    f1, G1 = bk.bk_forward(*args, **kw)
    r1 = bk_core.rows_recomputed()
    bk_core._DEBUG["row_units"] = 0
    torch.cuda.synchronize()
    return (f0, G0, r0), (f1, G1, r1)


def _inputs(B, N, seed, dtype=torch.float32, varying=False):
    g = torch.Generator().manual_seed(seed)
    a = (-2.0 + 0.5 * torch.randn(B, N, generator=g)).to(dtype)
    if varying:
        b = (1.0 + 0.1 * torch.randn(B, N - 1, generator=g)).to(dtype)
        c = (1.0 + 0.1 * torch.randn(B, N - 1, generator=g)).to(dtype)
    else:
        b = torch.ones(B, N - 1, dtype=dtype)
        c = torch.ones(B, N - 1, dtype=dtype)
    return a.cuda(), b.cuda(), c.cuda()


@pytest.mark.parametrize("N", [4 * CH, 4 * CH + 1, 5 * CH - 7, 6 * CH + 15, 7 * CH + 17, 8 * CH, 10 * CH - 4097 + 3, 13 * CH + 100])
@pytest.mark.parametrize("flag", [16, 32])
def test_units_equal_whole_rows_bitwise(bk, N, flag):
    from oracle import bk_oracle_c as C
    B = 133   # more than 128 rows: up to there long rows take the several-launch form for few rows instead
    a, b, c = _inputs(B, N, seed=N, varying=True)
    z = 0.3 + 0.9j
    (f0, G0, r0), (f1, G1, r1) = _both(bk, flag, a, b, c, z)
    assert r0 == 0 and r1 == 0
    assert torch.equal(torch.view_as_real(G0), torch.view_as_real(G1))
    assert torch.equal(f0, f1)
    ref = C.get_diag(a[:2].cpu().numpy(), b[:2].cpu().numpy(), c[:2].cpu().numpy(), z)
    # operator output = clipped get-diag; these inputs stay far from the clip
    assert float(np.max(np.abs(G1[:2].cpu().numpy() - ref) / np.abs(ref))) <= 1e-4


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float64])
def test_units_other_operand_types(bk, dtype):
    a, b, c = _inputs(6, 6 * CH + 33, seed=3, dtype=dtype, varying=True)   # (7 chunks: below the few-rows form)
    (f0, G0, r0), (f1, G1, r1) = _both(bk, 32, a, b, c, 1j)
    assert r0 == 0 and r1 == 0
    assert torch.equal(torch.view_as_real(G0), torch.view_as_real(G1))


def test_units_exact_phi_raw_and_clamped_features(bk):
    a, b, c = _inputs(131, 8 * CH - 5, seed=4, varying=True)
    for kw in (dict(phi_exact=True), dict(raw=True), dict(feature_clamp=0.25), dict(want_features=False)):
        (f0, G0, r0), (f1, G1, r1) = _both(bk, 32, a, b, c, 0.2 + 1.1j, **kw)
        assert r0 == 0 and r1 == 0, kw
        assert torch.equal(torch.view_as_real(G0), torch.view_as_real(G1)), kw
        if f0 is not None:
            assert torch.equal(f0, f1), kw


def test_units_more_rows_than_places(bk):
    """What the library does by itself: 1024 rows on 768 places, two units a row, no flag."""
    from bkcore import bk_core
    from oracle import bk_oracle_c as C
    B, N = 1024, 8 * CH
    a, b, c = _inputs(B, N, seed=5)
    z = torch.tensor(1j, dtype=torch.complex64, device="cuda")
    _, G = bk.bk_forward(a, b, c, z)
    assert bk_core.rows_recomputed() == 0
    torch.cuda.synchronize()
    st = bk_core._state_block(a.device, torch.cuda.current_stream().cuda_stream).tolist()
    assert st[9] > 0 and st[12] == 0          # the unit form ran (its epoch moved on) and no hand-over was given up
    _, G2 = bk.bk_forward(a, b, c, z)
    assert torch.equal(torch.view_as_real(G), torch.view_as_real(G2))
    rows = [0, 1, 255, 256, 767, 768, 1023]
    ref = C.forward_f32(a[rows].cpu().numpy(), b[rows].cpu().numpy(), c[rows].cpu().numpy(), 1j)[1]
    assert float(np.max(np.abs(G[rows].cpu().numpy() - ref) / np.abs(ref))) <= 1e-4
    # and a batch of some of the same rows run as whole rows: bit for bit
    rows = list(range(0, B, 5))
    bk_core._DEBUG["row_units"] = UNCUT
    _, Gs = bk.bk_forward(a[rows].contiguous(), b[rows].contiguous(), c[rows].contiguous(), z)
    bk_core._DEBUG["row_units"] = 0
    assert torch.equal(torch.view_as_real(Gs), torch.view_as_real(G[rows]))


def test_units_rows_that_fail_their_checks(bk):
    """One ill-conditioned row (left half only, right half only, across the cut): flagged by whichever unit sees it,
    recomputed by the second launch; its neighbours keep the unit form's values."""
    from bkcore import bk_core
    B, N = 130, 8 * CH
    a, b, c = _inputs(B, N, seed=6)
    g = torch.Generator().manual_seed(60)
    a[1, 100:400] = torch.randn(300, generator=g).cuda() * 0.05            # left unit
    a[3, N - 900:N - 500] = torch.randn(400, generator=g).cuda() * 0.05    # right unit
    a[4, 4 * CH - 20:4 * CH + 20] = torch.randn(40, generator=g).cuda() * 0.05  # across the cut
    (f0, G0, r0), (f1, G1, r1) = _both(bk, 16, a, b, c, 0.05j)
    assert r0 == r1 and r1 >= 3
    assert torch.equal(torch.view_as_real(G0), torch.view_as_real(G1))


def test_units_nan_rules(bk, golden):
    cs = golden["poisoned"]
    ap, bp, cp = (torch.from_numpy(np.ascontiguousarray(cs[k])).float() for k in ("a", "b", "c"))
    # the golden case is short; embed it at the start of long well-conditioned rows so that the units exist
    B, n0 = ap.shape
    N = 4 * CH + 50
    a, b, c = _inputs(B, N, seed=7)
    a[:, :n0] = ap.cuda()
    b[:, :n0 - 1] = bp.cuda()
    c[:, :n0 - 1] = cp.cuda()
    z = complex(*cs["z"])
    (f0, G0, r0), (f1, G1, r1) = _both(bk, 16, a, b, c, z)
    assert torch.equal(torch.view_as_real(G0), torch.view_as_real(G1))
    assert torch.equal(f0, f1)
    assert bool(torch.isfinite(torch.view_as_real(G1)).all())
    # NaN only in the off-diagonals of the right unit, clean diagonal: the NaN stays un-repaired as in whole rows
    a, b, c = _inputs(3, N, seed=8)
    b[1, N - 10] = float("nan")
    (f0, G0, r0), (f1, G1, r1) = _both(bk, 16, a, b, c, 1j)
    assert torch.equal(torch.view_as_real(G0).nan_to_num(7.0), torch.view_as_real(G1).nan_to_num(7.0))


@pytest.mark.parametrize("own_state", [False, True])
def test_units_graph_replay(bk, own_state):
    """A captured forward replays on new data.  own_state: the state block exists before the capture, so the epoch
    that tags the hand-over records lives on across replays and moves on in each; otherwise the block is created
    (and zeroed) inside the captured region, every replay starts from epoch 0, and what keeps the records of one
    replay from passing for those of the next is that their reader takes the tags down."""
    from bkcore import bk_core
    bk_core._DEBUG["keep_ws"] = False
    B, N = 8, 6 * CH + 9
    a, b, c = _inputs(B, N, seed=9)
    z = torch.tensor(1j, dtype=torch.complex64, device="cuda")
    bk_core._DEBUG["row_units"] = 32
    graph = torch.cuda.CUDAGraph()
    if own_state:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                bk.bk_forward(a, b, c, z)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=s):
            _, Gg = bk.bk_forward(a, b, c, z)
    else:
        with torch.cuda.graph(graph):
            _, Gg = bk.bk_forward(a, b, c, z)
    outs = []
    for k in range(4):
        a2, _, _ = _inputs(B, N, seed=90 + k)
        a.copy_(a2)
        graph.replay()
        torch.cuda.synchronize()
        outs.append((a.clone(), Gg.clone()))
    bk_core._DEBUG["row_units"] = UNCUT
    for a_k, G_k in outs:
        _, G_e = bk.bk_forward(a_k, b, c, z)
        assert torch.equal(torch.view_as_real(G_e), torch.view_as_real(G_k))
    bk_core._DEBUG["row_units"] = 0


def test_units_give_up_and_recover(bk):
    """BKCORE_FLAG_ROW_UNITS_MUTE: no unit publishes its record, so every left unit gives its wait up (50 ms), the
    launch raises the recompute-everything flag, and the call's last launch redoes every row on the two-scan path:
    the results are that path's (the operator's tolerance from the oracle), the state block is left clean, and the
    next call is an ordinary one."""
    import time
    from bkcore import bk_core
    from oracle import bk_oracle_c as C
    B, N = 20, 4 * CH + 77
    a, b, c = _inputs(B, N, seed=11, varying=True)
    z = 0.3 + 0.9j
    bk_core._DEBUG["row_units"] = 16 | 64
    torch.cuda.synchronize()
    t0 = time.time()
    _, G = bk.bk_forward(a, b, c, z)
    torch.cuda.synchronize()
    waited = time.time() - t0
    bk_core._DEBUG["row_units"] = 0
    assert 0.04 <= waited <= 5.0, waited
    st = bk_core._state_block(a.device, torch.cuda.current_stream().cuda_stream).tolist()
    # (words 6-8 belong to the opt-in adaptive path and keep what earlier calls on this stream left there)
    assert st[:6] == [0] * 6 and st[12] == 0
    ref = C.get_diag(a.cpu().numpy(), b.cpu().numpy(), c.cpu().numpy(), z)
    assert float(np.max(np.abs(G.cpu().numpy() - ref) / np.abs(ref))) <= 1e-4
    bk_core._DEBUG["scan_path"] = True          # every row from the scan kernels: what the recovery ran
    try:
        _, Gs = bk.bk_forward(a, b, c, z)
    finally:
        bk_core._DEBUG["scan_path"] = False
    # (the scan path's first launch is its fast form, the recovery its two-scan form: same tolerance, not bitwise)
    assert float((G - Gs).abs().max() / Gs.abs().max()) <= 1e-4
    (f0, G0, r0), (f1, G1, r1) = _both(bk, 16, a, b, c, z)  # and the units work again right after
    assert r0 == 0 and r1 == 0
    assert torch.equal(torch.view_as_real(G0), torch.view_as_real(G1))


@pytest.mark.parametrize("seed", range(6))
def test_units_by_themselves_random_shapes(bk, seed):
    """No flag: row counts in the band where the library takes units by itself (1.24 to 1.37 times the 768 places of an
    MI355X), lengths of 4 to 9 chunks with ragged ends, float32 or float64 operands.  Rows in units against the same
    rows run as whole rows: bit for bit."""
    from bkcore import bk_core
    rng = np.random.RandomState(1000 + seed)
    B = int(rng.randint(953, 1053))
    N = int(rng.randint(4 * CH - 50, 9 * CH))
    dtype = [torch.float32, torch.float32, torch.float64][seed % 3]
    a, b, c = _inputs(B, N, seed=2000 + seed, dtype=dtype, varying=bool(seed % 2))
    z = complex(0.1 * seed, 0.8 + 0.1 * seed)
    _, G = bk.bk_forward(a, b, c, z, want_features=False)
    assert bk_core.rows_recomputed() == 0
    torch.cuda.synchronize()
    if N > 4 * CH - 4096 + 1 and (N + CH - 1) // CH >= 4:
        st = bk_core._state_block(a.device, torch.cuda.current_stream().cuda_stream).tolist()
        assert st[9] > 0                       # a form with hand-overs ran (its epoch moved on)
    rows = sorted(set(rng.randint(0, B, size=200).tolist()) | {0, B - 1})
    bk_core._DEBUG["row_units"] = UNCUT
    _, Gs = bk.bk_forward(a[rows].contiguous(), b[rows].contiguous(), c[rows].contiguous(), z, want_features=False)
    bk_core._DEBUG["row_units"] = 0
    assert torch.equal(torch.view_as_real(Gs), torch.view_as_real(G[rows]))


def test_units_repeatable_under_other_work(bk):
    """The hand-over is timing dependent by nature (a workgroup polls for what another publishes): many calls, with
    and without another stream keeping the GPU busy, must give the bits of the first call, and no wait may be given up."""
    from bkcore import bk_core
    bk_core._DEBUG["keep_ws"] = False
    B, N = 1024, 8 * CH
    a, b, c = _inputs(B, N, seed=21)
    z = torch.tensor(1j, dtype=torch.complex64, device="cuda")
    _, G0 = bk.bk_forward(a, b, c, z, want_features=False)
    G0 = G0.clone()
    side = torch.cuda.Stream()
    x = torch.randn(64 << 20, device="cuda")
    bad = 0
    for it in range(120):
        if it % 3 == 0:
            with torch.cuda.stream(side):                 # competing memory traffic and workgroups
                y = x * 1.0001 + 0.5
                y2 = torch.cumsum(y[: 1 << 22], 0)
        _, G = bk.bk_forward(a, b, c, z, want_features=False)
        if it % 10 == 9:
            bad += int(not torch.equal(torch.view_as_real(G), torch.view_as_real(G0)))
    torch.cuda.synchronize()
    st = bk_core._state_block(a.device, torch.cuda.current_stream().cuda_stream).tolist()
    assert bad == 0 and st[12] == 0 and st[2] == 0


def test_units_on_two_streams_at_once(bk):
    """Two forwards of the headline shape on two streams at once, each in the form whose grid assumes the GPU to itself
    (each stream has its own state block and workspace): the bits of the serial results, and no hand-over given up (a
    given-up hand-over costs 50 ms)."""
    import time
    from bkcore import bk_core
    bk_core._DEBUG["keep_ws"] = False
    B, N = 1024, 8 * CH
    xs = [_inputs(B, N, seed=31), _inputs(B, N, seed=32)]
    z = torch.tensor(1j, dtype=torch.complex64, device="cuda")
    ref = [bk.bk_forward(*x, z)[1].clone() for x in xs]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for s, x in zip(streams, xs):
        with torch.cuda.stream(s):
            bk.bk_forward(*x, z)
    torch.cuda.synchronize()
    worst = 0.0
    for _ in range(40):
        outs = []
        t0 = time.perf_counter()
        for s, x in zip(streams, xs):
            with torch.cuda.stream(s):
                outs.append(bk.bk_forward(*x, z)[1])
        torch.cuda.synchronize()
        worst = max(worst, time.perf_counter() - t0)
        for o, r in zip(outs, ref):
            assert torch.equal(torch.view_as_real(o), torch.view_as_real(r))
    assert worst < 0.03, worst
