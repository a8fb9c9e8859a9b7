"""bkcore_forward_noalloc (scratch in a fixed device array of the library, nothing allocated by the call) against
bkcore_forward_clamped on the same inputs, through the C ABI: two calls in a row, the second with a smaller input;
then the size limit."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _call_pair(L, _lib, B, N, seed, poison=False):
    g = torch.Generator().manual_seed(seed)
    a = (-2.0 + 0.5 * torch.randn(B, N, generator=g)).cuda()
    b = (1.0 + 0.1 * torch.randn(B, N - 1, generator=g)).cuda()
    c = (1.0 + 0.1 * torch.randn(B, N - 1, generator=g)).cuda()
    if poison:                      # exercises the state block: NaN in he_diag and in b -> b's NaN is replaced
        a[0, 3] = float("nan")
        b[1, 5] = float("nan")
    stream = torch.cuda.current_stream().cuda_stream
    outs = []
    for which in ("ref", "noalloc"):
        G = torch.full((B, N), 7.0 + 7.0j, dtype=torch.complex64, device="cuda")
        F = torch.full((B, N, 2), 7.0, dtype=torch.float32, device="cuda")
        if which == "ref":
            nbytes = int(L.bkcore_workspace_bytes(B, N))
            ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device="cuda")
            st = torch.zeros(_lib.STATE_BYTES // 4, dtype=torch.int32, device="cuda")
            rc = L.bkcore_forward_clamped(a.data_ptr(), N, b.data_ptr(), N - 1, c.data_ptr(), N - 1, _lib.F32, None,
                                          0.3, 0.9, B, N, G.data_ptr(), F.data_ptr(), 5.0, ws.data_ptr(), nbytes,
                                          st.data_ptr(), 0, stream)
        else:
            rc = L.bkcore_forward_noalloc(a.data_ptr(), N, b.data_ptr(), N - 1, c.data_ptr(), N - 1, _lib.F32, None,
                                          0.3, 0.9, B, N, G.data_ptr(), F.data_ptr(), 5.0, 0, stream)
        assert rc == 0, (which, rc)
        torch.cuda.synchronize()
        outs.append((G.cpu().numpy(), F.cpu().numpy()))
    (G0, F0), (G1, F1) = outs
    assert np.array_equal(G0.view(np.uint32), G1.view(np.uint32))     # same kernels: bit for bit, NaN-safe
    assert np.array_equal(F0.view(np.uint32), F1.view(np.uint32))
    assert not np.any(G1 == 7.0 + 7.0j)


def test_noalloc_matches_forward_two_calls_in_a_row():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from bkcore import _lib
    L = _lib.load()
    _call_pair(L, _lib, 24, 9000, 1, poison=True)    # one workgroup per row, ragged last chunk, NaN corner
    _call_pair(L, _lib, 5, 700, 2)                   # smaller input straight after, same scratch
    _call_pair(L, _lib, 2, 70000, 3)                 # rows cut into segments (segment records in the scratch)
    _call_pair(L, _lib, 3, 300, 4, poison=True)


def test_noalloc_rejects_sizes_beyond_its_scratch():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from bkcore import _lib
    L = _lib.load()
    B, N = 1024, 32768
    assert int(L.bkcore_workspace_bytes(B, N)) > _lib.NOALLOC_MAX_WORKSPACE
    x = torch.zeros(8, device="cuda")
    rc = L.bkcore_forward_noalloc(x.data_ptr(), N, x.data_ptr(), N - 1, x.data_ptr(), N - 1, _lib.F32, None, 0.0, 1.0,
                                  B, N, x.data_ptr(), None, 0.0, 0, torch.cuda.current_stream().cuda_stream)
    assert rc == _lib.ETOOBIG
    assert b"BKCORE_NOALLOC_MAX_WORKSPACE" in L.bkcore_error_string(rc)


def test_noalloc_first_call_inside_graph_capture():
    """The callers this entry point is for include code under HIP graph capture.  In a fresh process (so that the
    call inside the capture is the first on the device and looks the scratch array's address up there): capture,
    replay on clean and NaN-poisoned inputs, compare bit for bit with bkcore_forward_clamped."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import os
    import subprocess
    import sys
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "noalloc_graph_child.py")
    r = subprocess.run([sys.executable, child], capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "bitwise equal on 3 replays" in r.stdout
