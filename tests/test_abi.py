"""CPU-side checks of the C-ABI shared library: it loads and exports every symbol include/bkcore.h declares.
No compute call is made here (no GPU in the build container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "bkcore.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bkcore_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_expected_entry_points():
    fns = declared_functions()
    for must in ("bkcore_forward", "bkcore_backward_ref", "bkcore_max_seq_len", "bkcore_error_string"):
        assert must in fns


def test_library_exports_every_declared_symbol():
    from bkcore import _lib
    assert os.path.exists(_lib.LIB_PATH), "build libbkcore.so first (python -c 'import __graft_entry__ as g; g.build()')"
    L = ctypes.CDLL(_lib.LIB_PATH)
    for fn in declared_functions():
        assert hasattr(L, fn), f"{fn} declared in include/bkcore.h but not exported by libbkcore.so"
    assert set(_lib.EXPORTS) <= set(declared_functions())


def test_host_side_argument_checks():
    # these return before any HIP call, so they run without a GPU
    from bkcore import _lib
    L = _lib.load()
    assert L.bkcore_abi_version() == _lib.ABI_VERSION
    assert L.bkcore_max_seq_len() >= 1 << 20
    assert L.bkcore_forward(None, 0, None, 0, None, 0, 0, None, 0.0, 1.0, 4, 16, None, None, None, 0, None, 0, None) == -1
    assert L.bkcore_forward(None, 0, None, 0, None, 0, 0, None, 0.0, 1.0, 0, 16, None, None, None, 0, None, 0, None) == 0
    assert L.bkcore_forward(None, 0, None, 0, None, 0, 0, None, 0.0, 1.0, -1, 16, None, None, None, 0, None, 0, None) == -1
    assert L.bkcore_workspace_bytes(1024, 32768) == 1024 * 4 + 1024 * 8 * 5 * 256 * 4
    assert L.bkcore_workspace_bytes(16, 1 << 20) == 256 + 16 * 256 * 5 * 256 * 4 + 16 * 64 * 32 * 4  # + segment records
    assert L.bkcore_workspace_bytes(0, 128) == 0
    # a problem that needs scratch is refused without it (checked before any launch)
    one = ctypes.c_void_p(16)
    assert L.bkcore_forward(one, 0, one, 0, one, 0, 0, None, 0.0, 1.0, 1, 1 << 20, one, None, None, 0, None, 0, None) == -4
    assert L.bkcore_backward_ref(None, None, None, 0.5, 0, None, None) == 0
    assert L.bkcore_backward_ref(None, None, None, 0.5, 8, None, None) == -1
    assert b"invalid" in L.bkcore_error_string(-1)
    assert b"sequence length" in L.bkcore_error_string(-2)


def test_operator_refuses_cpu_tensors():
    import torch
    import bkcore
    a = torch.zeros(2, 8)
    with pytest.raises(RuntimeError, match="ROCm device"):
        bkcore.BKCoreFunction.apply(a, a[:, :-1], a[:, :-1], 1j)


def test_reference_api_surface():
    # names the reference module exports (src/models/bk_core.py) exist with the same call shape
    import inspect
    import bkcore
    from bkcore.bk_core import BKCoreFunction
    assert BKCoreFunction.GRAD_BLEND == 0.5
    assert hasattr(BKCoreFunction, "USE_TRITON") and hasattr(BKCoreFunction, "USE_PARALLEL_SCAN")
    params = list(inspect.signature(BKCoreFunction.forward).parameters)
    assert params == ["ctx", "he_diag", "h0_super", "h0_sub", "z", "use_triton", "use_parallel_scan"]
    assert bkcore.get_triton_mode() is False


def test_launch_failure_reports_the_hip_code_from_every_translation_unit():
    """Without a GPU every launch fails (hipErrorNoDevice): each entry point must return BKCORE_ELAUNCH and leave the
    HIP code in bkcore_last_hip_error(), whichever translation unit it lives in.  (Nothing is computed: the launch is
    refused before any kernel runs.  With a GPU present the test has nothing to provoke and is skipped.)"""
    import torch
    from bkcore import _lib
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: launches succeed")
    L = _lib.load()
    buf = (ctypes.c_float * 256)()
    p = ctypes.addressof(buf)
    calls = {
        "bkcore_backward_ref (bkcore_kernels.hip)": lambda: L.bkcore_backward_ref(p, p, None, 0.5, 4, p, None),
        "bkcore_cumsum (bkcore_scan.hip)": lambda: L.bkcore_cumsum(p, p, 1, 16, 1, 0, None),
        "bkcore_layer_epilogue (bkcore_layer.hip)": lambda: L.bkcore_layer_epilogue(p, p, p, p, None, 2, 4, p, None),
    }
    for name, call in calls.items():
        rc = call()
        assert rc == _lib.ELAUNCH, name
        assert L.bkcore_last_hip_error() != 0, name


def test_forward_plan_policy():
    """bkcore_forward_plan: which form of the forward a shape takes (pure host logic; a GPU with 256 CUs assumed).
    Float32 operands hold 3 workgroups per CU, 16-bit operands 2."""
    from bkcore import _lib
    L = _lib.load()
    n = ctypes.c_int(0)

    def plan(B, N, dt=_lib.F32, flags=0):
        s = L.bkcore_forward_plan(B, N, dt, flags, 256, ctypes.byref(n))
        return s, n.value

    # pieces per row: 1 = whole rows, 2 = two units on a resident grid, chunks per row = one workgroup per chunk
    assert plan(1024, 32768) == (2, 2)              # the headline shape: 1024 rows on 768 places, two units a row
    assert plan(953, 32768) == (2, 2) and plan(1052, 32768) == (2, 2)    # ... the only band where units beat chunks
    assert plan(952, 32768) == (8, 2) and plan(1053, 32768) == (8, 2)
    assert plan(768, 32768) == (1, 2) and plan(730, 32768) == (1, 2)     # a full round of whole rows
    assert plan(729, 32768) == (8, 2) and plan(129, 32768) == (8, 2)     # fewer rows: chunks fill the GPU evenly
    assert plan(769, 32768) == (8, 2) and plan(1420, 32768) == (8, 2)    # a ragged second round: chunks
    assert plan(1421, 32768) == (1, 2)
    assert plan(1536, 32768) == (1, 2) and plan(2000, 32768) == (1, 2) and plan(8192, 32768) == (1, 2)
    assert plan(1024, 3 * 4096) == (1, 2)           # fewer than four chunks and more rows than places: whole rows
    assert plan(600, 3 * 4096) == (3, 2) and plan(600, 2 * 4096) == (1, 2)
    # 16-bit operands: whole rows (two per CU, 512 places) when their last round is full or nearly so, chunks otherwise
    assert plan(1024, 32768, _lib.BF16) == (1, 2) and plan(512, 32768, _lib.BF16) == (1, 2)
    assert plan(980, 32768, _lib.BF16) == (1, 2) and plan(1536, 32768, _lib.F16) == (1, 2)
    assert plan(600, 32768, _lib.BF16) == (8, 2) and plan(1080, 32768, _lib.BF16) == (8, 2)
    assert plan(900, 32768, _lib.BF16) == (8, 2) and plan(400, 32768, _lib.BF16) == (8, 2)
    assert plan(1024, 32768, flags=_lib.FLAG_ADAPTIVE) == (1, 2)
    assert plan(1024, 1 << 18) == (1, 2)            # rows beyond 49152 elements run in double, two workgroups per CU
    assert plan(600, 1 << 18) == (64, 2)
    assert plan(16, 1 << 20) == (256, 2)            # few long rows: one workgroup per chunk, handing over in the launch
    s, l = plan(16, 1 << 20, flags=_lib.FLAG_SCAN_PATH)
    assert s > 1 and l == 4                         # scan form: aggregates / combination / outputs / rows to recompute
    assert plan(100, 4 * 4096, flags=_lib.FLAG_ROW_UNITS) == (2, 2)
    assert plan(1024, 32768, flags=_lib.FLAG_WHOLE_ROWS) == (8, 2)      # no resident grid (data-parallel training): chunks
    assert plan(16, 1 << 20, flags=_lib.FLAG_WHOLE_ROWS) == (256, 2)
    assert L.bkcore_forward_plan(0, 5, _lib.F32, 0, 256, None) == _lib.EINVAL
