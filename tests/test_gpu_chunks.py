"""Few long rows (B <= 128, N >= 8 chunks): one workgroup per chunk, the running product handed from chunk to chunk
inside ONE launch through tagged records (every 64th chunk also publishes the running product, so which records a
chunk combines is fixed and the result does not depend on timing).  The reference's counterpart is the chunk loop of
its Triton scan with the carry re-loaded from global memory (src/kernels/bk_scan.py:194-386, :380-384).
Checked here: parity with the oracle across anchor boundaries and operand types, two launches per call, bitwise
repeatability (also with other work on the GPU and in graph replay), records of an earlier call that must not pass
for this call's, and the give-up path."""
import ctypes
import time

import numpy as np
import pytest
import torch

from conftest import max_rel_err

pytestmark = pytest.mark.gpu

CH = 4096  # elements per chunk of the ratio-form kernel
TOL = 1e-4


@pytest.fixture()
def bk():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import bkcore
    from bkcore import bk_core
    yield bkcore
    bk_core._DEBUG["keep_ws"] = False
    bk_core._DEBUG["row_units"] = 0
    bk_core._DEBUG["last_ws"] = None


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


def _fresh(bk, *args, **kw):
    """Forward with a workspace of its own (uninitialised memory); returns G and the number of rows recomputed."""
    from bkcore import bk_core
    bk_core._DEBUG["keep_ws"] = True
    try:
        _, G = bk.bk_forward(*args, **kw)
        torch.cuda.synchronize()
        return G, bk_core.rows_recomputed()
    finally:
        bk_core._DEBUG["keep_ws"] = False
        bk_core._DEBUG["last_ws"] = None


def test_plan_is_two_launches(bk):
    from bkcore import _lib
    L = _lib.load()
    n = ctypes.c_int(0)
    assert L.bkcore_forward_plan(16, 1 << 20, _lib.F32, 0, 0, ctypes.byref(n)) == 256 and n.value == 2
    assert L.bkcore_forward_plan(128, 8 * CH, _lib.BF16, 0, 0, ctypes.byref(n)) == 8 and n.value == 2
    assert L.bkcore_forward_plan(3, 8 * CH + 1, _lib.F64, _lib.FLAG_PHI_EXACT, 0, ctypes.byref(n)) == 9 and n.value == 2


@pytest.mark.parametrize("B,N,dtype,conv", [
    (2, 63 * CH + 11, torch.float32, "reference"),     # double chains (N > 49152), no anchor yet
    (2, 64 * CH, torch.float32, "reference"),          # the first anchor is the row's last chunk: nothing reads it
    (3, 64 * CH + 1, torch.float32, "exact"),          # one chunk reads the anchor alone
    (2, 65 * CH + 4000, torch.bfloat16, "reference"),
    (1, 128 * CH + 17, torch.float64, "reference"),    # two anchors, a ragged last chunk
    (2, 130 * CH - 5, torch.float16, "reference"),
    (128, 8 * CH, torch.float32, "reference"),         # the most rows and the fewest chunks that take this form
    (7, 32 * CH, torch.float32, "exact"),
    (7, 12 * CH, torch.float32, "exact"),              # float32 chains at their longest
])
def test_parity_across_anchor_boundaries(bk, B, N, dtype, conv):
    from oracle import bk_oracle_c as C
    a, b, c = _inputs(B, N, seed=N % 1000 + B, dtype=dtype, varying=True)
    z = 0.3 + 0.9j
    G, redone = _fresh(bk, a, b, c, z, phi_exact=(conv == "exact"))
    assert redone == 0
    rows = sorted({0, B - 1})
    ref = C.get_diag(a[rows].double().cpu().numpy(), b[rows].double().cpu().numpy(), c[rows].double().cpu().numpy(), z,
                     phi_exact=(conv == "exact"))
    assert max_rel_err(G[rows].cpu().numpy(), ref) <= TOL
    G2, _ = _fresh(bk, a, b, c, z, phi_exact=(conv == "exact"))
    assert torch.equal(torch.view_as_real(G), torch.view_as_real(G2))


def test_records_of_an_earlier_call_do_not_pass(bk):
    """Same workspace, same shape, state block zeroed between the calls (what the captured memset of a replayed graph
    does): the epoch that tags the records repeats, so the second call would find the first call's records complete
    -- were they not taken down by the first call's last launch.  Both forms that hand over inside a launch."""
    from bkcore import bk_core
    for B, N, flag in ((4, 70 * CH + 9, 0), (6, 9 * CH, 0), (133, 6 * CH + 5, 16)):
        bk_core._DEBUG["row_units"] = flag
        a1, b1, c1 = _inputs(B, N, seed=31, varying=True)
        a2, b2, c2 = _inputs(B, N, seed=32, varying=True)
        z = 0.3 + 0.9j
        want, _ = _fresh(bk, a2, b2, c2, z)
        st = bk_core._state_block(a1.device, torch.cuda.current_stream().cuda_stream)
        for _ in range(2):
            bk.bk_forward(a1, b1, c1, z)                 # shared per-stream workspace: records of (a1, b1, c1)
            st.zero_()
            _, got = bk.bk_forward(a2, b2, c2, z)
            torch.cuda.synchronize()
            assert torch.equal(torch.view_as_real(got), torch.view_as_real(want)), (B, N, flag)
        bk_core._DEBUG["row_units"] = 0


def test_give_up_and_recover(bk):
    """BKCORE_FLAG_ROW_UNITS_MUTE: no chunk publishes anything.  The first chunk that waits gives up after 50 ms and
    raises the recompute-everything flag, which ends every other wait at once; the call's last launch redoes every
    row on the two-scan path, leaves the state block clean, and the next call is an ordinary one."""
    from bkcore import bk_core
    from oracle import bk_oracle_c as C
    B, N = 5, 40 * CH + 77
    a, b, c = _inputs(B, N, seed=41, varying=True)
    z = 0.3 + 0.9j
    want, redone = _fresh(bk, a, b, c, z)
    assert redone == 0
    bk_core._DEBUG["row_units"] = 64
    torch.cuda.synchronize()
    t0 = time.time()
    _, G = bk.bk_forward(a, b, c, z)
    torch.cuda.synchronize()
    waited = time.time() - t0
    bk_core._DEBUG["row_units"] = 0
    assert 0.04 <= waited <= 2.0, waited              # one timeout, not one per chunk
    st = bk_core._state_block(a.device, torch.cuda.current_stream().cuda_stream).tolist()
    assert st[:6] == [0] * 6 and st[12] == 0
    ref = C.get_diag(a.cpu().numpy(), b.cpu().numpy(), c.cpu().numpy(), z)
    assert max_rel_err(G.cpu().numpy(), ref) <= TOL
    _, G2 = bk.bk_forward(a, b, c, z)
    assert torch.equal(torch.view_as_real(G2), torch.view_as_real(want))


def test_rows_that_fail_their_checks(bk):
    """One ill-conditioned stretch inside a chunk and one across a chunk boundary: those rows are flagged by whichever
    chunk sees them and recomputed by the second launch; the other rows keep their values bit for bit."""
    B, N = 6, 12 * CH + 100
    a, b, c = _inputs(B, N, seed=51)
    z = 1j
    clean, r0 = _fresh(bk, a, b, c, z)
    assert r0 == 0
    # small diagonal under large off-diagonals: |k| / |r|^2 ~ 0.7, the 16-step warm-up cannot forget its start
    for row, lo, hi in ((1, 5 * CH + 100, 5 * CH + 400), (4, 7 * CH - 20, 7 * CH + 20)):
        a[row, lo:hi] = 0.0
        b[row, lo:hi] = 3.0
        c[row, lo:hi] = 3.0
    G, r1 = _fresh(bk, a, b, c, z)
    assert r1 == 2
    keep = [0, 2, 3, 5]
    assert torch.equal(torch.view_as_real(G[keep]), torch.view_as_real(clean[keep]))
    assert bool(torch.isfinite(torch.view_as_real(G)).all())


def test_nan_rules(bk, golden):
    """The reference's sanitiser (bk_core.py:256-264) across chunks: a NaN in he_diag of one chunk repairs the NaNs of
    b and c everywhere -- which only the second launch can know."""
    cs = golden["poisoned"]
    ap, bp, cp = (torch.from_numpy(np.ascontiguousarray(cs[k])).float() for k in ("a", "b", "c"))
    Bp, n0 = ap.shape
    N = 9 * CH + 50
    a, b, c = _inputs(Bp, N, seed=61)
    a[:, :n0] = ap.cuda()
    b[:, :n0 - 1] = bp.cuda()
    c[:, :n0 - 1] = cp.cuda()
    b[0, N - 10] = float("nan")                       # far from the poisoned diagonal, in another chunk
    z = complex(*cs["z"])
    G, _ = _fresh(bk, a, b, c, z)
    assert bool(torch.isfinite(torch.view_as_real(G)).all())
    big = 131                                         # the same rows inside a batch that runs as whole rows
    A = a[:1].repeat(big, 1); Bm = b[:1].repeat(big, 1); Cm = c[:1].repeat(big, 1)
    A[:Bp], Bm[:Bp], Cm[:Bp] = a, b, c
    Gw, _ = _fresh(bk, A, Bm, Cm, z)
    assert torch.equal(torch.view_as_real(Gw[:Bp]), torch.view_as_real(G))   # both recomputed by the same kernel


def test_graph_replay(bk):
    """A captured forward of this form replays on new data; the state block is created (and zeroed) inside the
    captured region, so every replay runs under the same epoch."""
    B, N = 8, 20 * CH + 9
    a, b, c = _inputs(B, N, seed=71)
    z = torch.tensor(1j, dtype=torch.complex64, device="cuda")
    s = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        _, Gg = bk.bk_forward(a, b, c, z)
    outs = []
    for k in range(4):
        a2, _, _ = _inputs(B, N, seed=710 + k)
        a.copy_(a2)
        graph.replay()
        torch.cuda.synchronize()
        outs.append((a.clone(), Gg.clone()))
    for a_k, G_k in outs:
        G_e, _ = _fresh(bk, a_k, b, c, z)
        assert torch.equal(torch.view_as_real(G_e), torch.view_as_real(G_k))


def test_repeatable_under_other_work(bk):
    """The hand-over is timing dependent by nature; which records are combined is not.  Many calls at the
    long-context shape, with another stream keeping the GPU busy: the bits of the first call, no wait given up."""
    from bkcore import bk_core
    B, N = 16, 1 << 20
    a, b, c = _inputs(B, N, seed=81)
    z = torch.tensor(1j, dtype=torch.complex64, device="cuda")
    _, G0 = bk.bk_forward(a, b, c, z, want_features=False)
    G0 = G0.clone()
    side = torch.cuda.Stream()
    x = torch.randn(64 << 20, device="cuda")
    bad = 0
    for it in range(60):
        if it % 3 == 0:
            with torch.cuda.stream(side):
                y = x * 1.0001 + 0.5
                y2 = torch.cumsum(y[: 1 << 22], 0)
        _, G = bk.bk_forward(a, b, c, z, want_features=False)
        if it % 6 == 5:
            bad += int(not torch.equal(torch.view_as_real(G), torch.view_as_real(G0)))
    torch.cuda.synchronize()
    st = bk_core._state_block(a.device, torch.cuda.current_stream().cuda_stream).tolist()
    assert bad == 0 and st[12] == 0 and st[2] == 0


@pytest.mark.parametrize("seed", range(6))
def test_chunks_by_themselves_random_shapes(bk, seed):
    """No flag: row counts at which whole rows would leave the GPU unevenly filled (fewer rows than 0.95 of its 768
    places, or between one and 1.85 rounds of them outside the band where units are ahead), 3 to 9 chunks with ragged
    ends, any operand type.  Parity with the oracle on a row subset, no row recomputed, the same bits on a second call,
    and within 5e-5 of the same rows run as whole rows (the running product is put together in another order)."""
    from bkcore import bk_core
    from oracle import bk_oracle_c as C
    rng = np.random.RandomState(3000 + seed)
    B = int([rng.randint(129, 730), rng.randint(769, 953), rng.randint(1053, 1421)][seed % 3])
    N = int(rng.randint(3 * CH - 50, 9 * CH)) if B < 730 else int(rng.randint(4 * CH - 50, 9 * CH))
    if (N + CH - 1) // CH < (3 if B < 730 else 4):
        N += CH
    dtype = [torch.float32, torch.bfloat16, torch.float16, torch.float64][seed % 4]
    a, b, c = _inputs(B, N, seed=4000 + seed, dtype=dtype, varying=bool(seed % 2))
    z = complex(0.1 * seed, 0.8 + 0.1 * seed)
    G, redone = _fresh(bk, a, b, c, z)
    assert redone == 0
    st = bk_core._state_block(a.device, torch.cuda.current_stream().cuda_stream).tolist()
    assert st[9] > 0 and st[12] == 0           # a form with hand-overs ran and none was given up
    G2, _ = _fresh(bk, a, b, c, z)
    assert torch.equal(torch.view_as_real(G), torch.view_as_real(G2))
    rows = sorted(set(rng.randint(0, B, size=6).tolist()) | {0, B - 1})
    ref = C.get_diag(a[rows].double().cpu().numpy(), b[rows].double().cpu().numpy(), c[rows].double().cpu().numpy(), z)
    assert max_rel_err(G[rows].cpu().numpy(), ref) <= TOL
    bk_core._DEBUG["row_units"] = 256          # BKCORE_FLAG_ROWS_UNCUT
    try:
        Gw, _ = _fresh(bk, a, b, c, z)
    finally:
        bk_core._DEBUG["row_units"] = 0
    assert max_rel_err(G.cpu().numpy(), Gw.cpu().numpy()) <= 5e-5
