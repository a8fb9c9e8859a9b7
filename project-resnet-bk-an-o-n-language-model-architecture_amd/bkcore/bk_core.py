"""BK-Core operator for MI355X: drop-in for the reference's ``src/models/bk_core.py``.

Same names, argument meaning and return values as the reference:

  * ``get_tridiagonal_inverse_diagonal(a, b, c, z)``          reference bk_core.py:27-210
  * ``vmapped_get_diag(a, b, c, z)``                           reference bk_core.py:214-216
  * ``BKCoreFunction.apply(he_diag, h0_super, h0_sub, z, use_triton=None, use_parallel_scan=None)
        -> (features (B,N,2) float32, G_ii (B,N) complex64)``  reference bk_core.py:219-385
  * ``set_triton_mode`` / ``get_triton_mode``                  reference bk_core.py:389-433

The arithmetic runs in hand-written HIP kernels reached through the C ABI of ``libbkcore.so``
(``include/bkcore.h``).  The numerics are those of the reference's sequential log-scaled path
(what ``apply(..., False, False)`` computes there); the ``use_triton`` / ``use_parallel_scan``
arguments and class attributes are accepted and ignored -- there is exactly one back-end.
Tensors must live on a ROCm device; there is no CPU path.
"""
from __future__ import annotations

import ctypes
import logging
import warnings

import torch

from . import _lib

logger = logging.getLogger(__name__)

_DTYPES = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16, torch.float16: _lib.F16, torch.float64: _lib.F64}
_DEBUG = {"keep_ws": False, "last_ws": None, "last_B": 0, "scan_path": False, "row_units": 0}  # diagnostics: see rows_recomputed()
_state = {}  # (device index, stream handle) -> persistent int32 state block for the NaN-sanitiser corner
_ws = {}     # (device index, stream handle) -> workspace tensor, grown on demand and reused by every call on the stream
_ws_bytes = {}  # (B, N) -> bkcore_workspace_bytes(B, N)
_ws_ana = {}    # the same two for the analytic backward
_ws_ana_bytes = {}
_ws_retired = []  # workspaces that a larger one superseded: kept, because a captured graph may still point at them


def _state_block(device: torch.device, stream: int) -> torch.Tensor:
    key = (device.index, stream)
    blk = _state.get(key)
    if blk is None:
        blk = torch.zeros(_lib.STATE_BYTES // 4, dtype=torch.int32, device=device)
        _state[key] = blk
    return blk


def _whole_rows_only() -> bool:
    """BKCoreFunction.ROW_UNITS: "auto" keeps the library's choice except in a process that is one rank of a
    torch.distributed job -- there a collective of another stream (the gradient all-reduce of a data-parallel backward)
    may hold compute units while the forward runs, and the form that cuts rows into units handing over inside one launch
    needs its whole grid resident at once; "never" / "always" force the answer."""
    mode = BKCoreFunction.ROW_UNITS
    if mode == "never":
        return True
    if mode == "always":
        return False
    dist = torch.distributed
    return bool(dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)


def _cached_workspace(cache: dict, device: torch.device, stream: int, need: int) -> torch.Tensor:
    """One buffer per (device, stream): calls on one stream execute in order, so it serves them all (every call on the
    stream shares it; it is written before it is read by each).  It only ever grows, and a buffer that a larger one
    supersedes is kept rather than freed: a HIP graph captured on this stream earlier holds its address.  While the
    stream is being captured the call gets a buffer of its own from the graph's memory pool instead -- the cached one
    belongs to eager calls, which may run concurrently with replays."""
    if torch.cuda.is_current_stream_capturing():
        return torch.empty(max(need, 256), dtype=torch.uint8, device=device)
    key = (device.index, stream)
    ws = cache.get(key)
    if ws is None or ws.numel() < need:
        if ws is not None:
            _ws_retired.append(ws)
        ws = torch.empty(max(need, 256), dtype=torch.uint8, device=device)
        cache[key] = ws
    return ws


def _workspace(L, device: torch.device, stream: int, B: int, N: int):
    """(tensor, bytes needed) for the forward.  The size query is cached per shape."""
    need = _ws_bytes.get((B, N))
    if need is None:
        need = int(L.bkcore_workspace_bytes(B, N))
        _ws_bytes[(B, N)] = need
    return _cached_workspace(_ws, device, stream, need), need


class _on_device:
    """`with _on_device(dev):` makes dev current for the launch; costs nothing when it already is (the usual case),
    unlike torch.cuda.device."""
    __slots__ = ("idx", "prev")

    def __init__(self, dev: torch.device):
        self.idx = dev.index
        self.prev = -1

    def __enter__(self):
        cur = torch.cuda.current_device()
        if cur != self.idx:
            self.prev = cur
            torch.cuda.set_device(self.idx)

    def __exit__(self, *exc):
        if self.prev >= 0:
            torch.cuda.set_device(self.prev)
        return False


def _rows(x: torch.Tensor, n_inner: int, name: str):
    """(tensor kept alive, data_ptr, leading dimension in elements) for a (B, n_inner) operand with unit inner
    stride.  A broadcast batch dimension (stride 0, as produced by expand()) is passed through as ld = 0."""
    if x.dim() != 2 or x.shape[1] != n_inner:
        raise ValueError(f"bkcore: {name} must have shape (B, {n_inner}), got {tuple(x.shape)}")
    if (n_inner > 1 and x.stride(1) != 1) or x.stride(0) < 0:
        x = x.contiguous()
    ld = x.stride(0) if x.shape[0] > 1 else 0
    return x, x.data_ptr(), ld


def _check_batch(B: int, b: torch.Tensor, c: torch.Tensor):
    """b and c hold one row per row of a, or a single row that is broadcast (leading dimension 0); anything else
    would make the kernels read past the operand (the reference's vmap raises on such shapes too)."""
    for name, x in (("h0_super", b), ("h0_sub", c)):
        if x.dim() == 2 and x.shape[0] not in (1, B):
            raise ValueError(f"bkcore: {name} has {x.shape[0]} rows for a batch of {B} (expected {B} or 1)")


def _prep_inputs(a, b, c):
    if a.is_complex() or b.is_complex() or c.is_complex():
        # as in the reference, whose input clamp (bk_core.py:256-258) is not defined for complex tensors
        raise TypeError("bkcore: the operator takes real he_diag / h0_super / h0_sub (its input clamp is not defined "
                        "for complex tensors); vmapped_get_diag / get_tridiagonal_inverse_diagonal accept complex ones")
    if not a.is_cuda:
        raise RuntimeError("bkcore: tensors must be on a ROCm device (no CPU path; use the oracle for CPU checks)")
    dt = a.dtype
    if dt not in _DTYPES:
        a = a.float()
        dt = torch.float32
    if b.dtype != dt:
        b = b.to(dt)
    if c.dtype != dt:
        c = c.to(dt)
    if b.device != a.device or c.device != a.device:
        raise RuntimeError("bkcore: a, b, c must be on the same device")
    return a, b, c, _DTYPES[dt]


def _prep_z(z, device):
    """-> (keepalive tensor or None, device pointer or None, re, im)"""
    if isinstance(z, (int, float, complex)):
        zc = complex(z)
        return None, None, float(zc.real), float(zc.imag)
    if not torch.is_tensor(z):
        raise TypeError(f"bkcore: z must be a python scalar or a 0-dim tensor, got {type(z)}")
    if z.numel() != 1:
        raise ValueError("bkcore: z must hold exactly one value")
    if z.is_cuda:
        # a complex64 element is the (re, im) float pair the kernels read: no view, no copy, no host sync
        # (a lazily conjugated or negated tensor keeps its bits un-conjugated in storage: materialise it first)
        if z.is_conj() or z.is_neg():
            z = z.resolve_conj().resolve_neg()
        if z.dtype == torch.complex64 and z.device == device:
            return z, z.data_ptr(), 0.0, 0.0
        zz = z.detach().reshape(()).to(device=device, dtype=torch.complex64)
        return zz, zz.data_ptr(), 0.0, 0.0
    zc = complex(z.detach().reshape(()).item())
    return None, None, float(zc.real), float(zc.imag)


def bk_forward(a, b, c, z, raw: bool = False, phi_exact: bool = False, want_features: bool = True,
               feature_clamp=None, share_storage: bool = False, want_states: bool = False):
    """Launch the HIP forward.  a (B,N), b/c (B,N-1) real tensors on the GPU.  Returns (features|None, G).
    feature_clamp: the features come out clamped to +-feature_clamp (G does not).
    share_storage: without a clamp the features are bit for bit the (re, im) pairs of G (bk_core.py:296-298), so they
    are returned as the float32 view of G instead of a second 8-byte-per-element write.
    want_states: returns (features|None, G, states|None) -- the boundary states of the ratio chains for
    bk_backward_analytic (a float32 tensor of bkcore_boundary_state_bytes(B, N) bytes, or None when this call's path does
    not keep them)."""
    L = _lib.load()
    a, b, c, dt = _prep_inputs(a, b, c)
    B, N = a.shape
    nb = max(N - 1, 0)
    G = torch.empty((B, N), dtype=torch.complex64, device=a.device)
    if want_features and share_storage and feature_clamp is None:
        feats_out = torch.view_as_real(G)
        feats = None                                   # nothing for the kernel to write
    else:
        feats = torch.empty((B, N, 2), dtype=torch.float32, device=a.device) if want_features else None
        feats_out = feats
    if B == 0 or N == 0:
        return (feats_out, G, None) if want_states else (feats_out, G)
    _check_batch(B, b, c)
    a, pa, lda = _rows(a, N, "a")
    b, pb, ldb = _rows(b, nb, "b")
    c, pc, ldc = _rows(c, nb, "c")
    dev = a.device
    states = None
    with _on_device(dev):
        stream = torch.cuda.current_stream().cuda_stream
        zkeep, pz, zre, zim = _prep_z(z, dev)
        state = _state_block(dev, stream)
        flags = (_lib.FLAG_RAW if raw else 0) | (_lib.FLAG_PHI_EXACT if phi_exact else 0) \
            | (_lib.FLAG_SCAN_PATH if _DEBUG["scan_path"] else 0) \
            | (_lib.FLAG_ADAPTIVE if BKCoreFunction.ADAPTIVE_PATH else 0) | _DEBUG["row_units"] \
            | (_lib.FLAG_WHOLE_ROWS if _whole_rows_only() else 0)
        ws, ws_bytes = _workspace(L, dev, stream, B, N)
        if _DEBUG["keep_ws"]:       # diagnostics read the workspace after the call: give this call its own
            ws = torch.empty(max(ws_bytes, 256), dtype=torch.uint8, device=dev)
        fc = 0.0
        if feature_clamp is not None:
            fc = float(feature_clamp)
            if not fc > 0.0:
                raise ValueError("bkcore: feature_clamp must be positive")
        if want_states and dt == _lib.F32:
            st_bytes = int(L.bkcore_boundary_state_bytes(B, N))   # 16 bytes per block of 8, rows padded
            states = torch.empty((B, st_bytes // (16 * B), 4), dtype=torch.float32, device=dev)
            written = ctypes.c_int(0)
            rc = L.bkcore_forward_states(pa, lda, pb if nb else None, ldb, pc if nb else None, ldc, dt, pz, zre, zim,
                                         B, N, G.data_ptr(), feats.data_ptr() if feats is not None else None, fc,
                                         states.data_ptr(), st_bytes, ctypes.byref(written), ws.data_ptr(), ws_bytes,
                                         state.data_ptr(), flags, stream)
            if not written.value:
                states = None
        else:
            rc = L.bkcore_forward_clamped(pa, lda, pb if nb else None, ldb, pc if nb else None, ldc, dt, pz, zre, zim,
                                          B, N, G.data_ptr(), feats.data_ptr() if feats is not None else None, fc,
                                          ws.data_ptr(), ws_bytes, state.data_ptr(), flags, stream)
    _lib.check(rc, "bkcore_forward")
    del zkeep
    if _DEBUG["keep_ws"]:
        _DEBUG["last_ws"], _DEBUG["last_B"] = ws, B
    return (feats_out, G, states) if want_states else (feats_out, G)


def rows_recomputed():
    """Diagnostics: number of rows of the last forward (run with _DEBUG['keep_ws'] = True) whose fast first pass
    failed its check and that the second launch recomputed on the two-scan path."""
    ws, B = _DEBUG["last_ws"], _DEBUG["last_B"]
    if ws is None:
        return None
    return int(ws[: 4 * B].view(torch.int32).ne(0).sum().item())


_NONFINITE_MSG = "BKCoreFunction backward: Incoming 'grad_output_features' contains NaN or Inf."
_nonfinite = {}     # device -> [device flag (int32), pinned host copy, event of the last read-back or None]


def _nonfinite_slot(device):
    slot = _nonfinite.get(device)
    if slot is None:
        slot = [torch.zeros(1, dtype=torch.int32, device=device), torch.zeros(1, dtype=torch.int32).pin_memory(), None]
        _nonfinite[device] = slot
    return slot


def poll_nonfinite_grad_warning(wait: bool = False) -> bool:
    """With ``BKCoreFunction.WARN_NONFINITE_GRAD = "lazy"`` the backward kernel raises a device-side flag when the
    incoming gradient holds a NaN or an infinity, and the flag travels to the host by an asynchronous copy.  This
    function (also called at the start of every later backward) emits the reference's warning (bk_core.py:321-325)
    for every read-back that has completed -- without waiting, unless ``wait`` -- and returns whether it warned."""
    warned = False
    for slot in _nonfinite.values():
        ev = slot[2]
        if ev is None:
            continue
        if wait:
            ev.synchronize()
        elif not ev.query():
            continue
        slot[2] = None
        if int(slot[1][0]) != 0:
            slot[1][0] = 0
            warnings.warn(_NONFINITE_MSG, RuntimeWarning)
            logger.warning(_NONFINITE_MSG)
            warned = True
    return warned


def bk_backward_reference(G, grad_features, grad_G, alpha: float, flag_nonfinite: bool = False):
    """Launch the reference-mode backward kernel.  Returns grad_he_diag float32 with G's shape.
    ``flag_nonfinite``: report a NaN / infinity in grad_features through poll_nonfinite_grad_warning()."""
    L = _lib.load()
    G = G.contiguous()
    out = torch.empty(G.shape, dtype=torch.float32, device=G.device)
    n = G.numel()
    if n == 0:
        return out
    gf = None
    if grad_features is not None:
        gf = grad_features
        if gf.dtype != torch.float32:
            gf = gf.float()
        gf = gf.contiguous()
    gg = None
    if grad_G is not None:
        gg = grad_G
        if gg.dtype != torch.complex64:
            gg = gg.to(torch.complex64)
        gg = gg.contiguous()
    with _on_device(G.device):
        stream = torch.cuda.current_stream().cuda_stream
        if flag_nonfinite and gf is not None:
            poll_nonfinite_grad_warning()
            slot = _nonfinite_slot(G.device)
            if slot[2] is not None:          # the previous read-back has not landed: let it, the flag is sticky
                slot[2].synchronize()
                poll_nonfinite_grad_warning()
            rc = L.bkcore_backward_ref_flagged(G.data_ptr(), gf.data_ptr(), gg.data_ptr() if gg is not None else None,
                                               float(alpha), n, out.data_ptr(), slot[0].data_ptr(), stream)
            slot[1].copy_(slot[0], non_blocking=True)
            slot[0].zero_()
            ev = torch.cuda.Event()
            ev.record()
            slot[2] = ev
        else:
            rc = L.bkcore_backward_ref(G.data_ptr(), gf.data_ptr() if gf is not None else None,
                                       gg.data_ptr() if gg is not None else None, float(alpha), n, out.data_ptr(),
                                       stream)
    _lib.check(rc, "bkcore_backward_ref")
    return out



def layer_prologue(v_pre, h0_diag, v_max: float, out_dtype=torch.bfloat16):
    """he_diag = h0_diag + v_max * tanh(nan_to_num(v_pre, 0, +-v_max) / v_max), cast to ``out_dtype``, in one launch
    (the elementwise chain of reference src/models/resnet_bk.py:139-141,158,165).  v_pre (B, N) float32 on the GPU,
    h0_diag N floats (any shape with N elements)."""
    L = _lib.load()
    if not v_pre.is_cuda:
        raise RuntimeError("bkcore: tensors must be on a ROCm device (no CPU path)")
    if out_dtype not in (torch.bfloat16, torch.float32):
        raise TypeError("bkcore: layer_prologue writes bfloat16 or float32")
    if v_pre.dim() != 2:
        raise ValueError(f"bkcore: v_pre must have shape (B, N), got {tuple(v_pre.shape)}")
    B, N = v_pre.shape
    v = v_pre if v_pre.dtype == torch.float32 else v_pre.float()
    if N > 1 and v.stride(1) != 1 or v.stride(0) < 0:
        v = v.contiguous()
    h0 = h0_diag.reshape(-1).to(device=v.device, dtype=torch.float32).contiguous()
    if h0.numel() != N:
        raise ValueError(f"bkcore: h0_diag must hold N = {N} values, got {h0.numel()}")
    out = torch.empty((B, N), dtype=out_dtype, device=v.device)
    if B == 0 or N == 0:
        return out
    with _on_device(v.device):
        stream = torch.cuda.current_stream().cuda_stream
        rc = L.bkcore_layer_prologue(v.data_ptr(), v.stride(0) if B > 1 else N, h0.data_ptr(), float(v_max), B, N,
                                     out.data_ptr(), _lib.BF16 if out_dtype == torch.bfloat16 else _lib.F32, stream)
    _lib.check(rc, "bkcore_layer_prologue")
    return out


def layer_backward_reference(G, grad_features, grad_G, alpha: float, feature_clamp, v_pre, v_max: float,
                             round_bf16: bool = True):
    """Gradient with respect to v_pre through clamp(features) -> operator (reference-mode backward) -> bfloat16 cast
    -> tanh saturation -> nan_to_num, in one launch.  Returns float32 with v_pre's shape."""
    L = _lib.load()
    G = G.contiguous()
    v = v_pre.float().contiguous()
    if v.numel() != G.numel():
        raise ValueError("bkcore: v_pre and G must hold the same number of elements")
    out = torch.empty(v.shape, dtype=torch.float32, device=G.device)
    n = G.numel()
    if n == 0:
        return out
    gf = grad_features.float().contiguous() if grad_features is not None else None
    gg = grad_G.to(torch.complex64).contiguous() if grad_G is not None else None
    with _on_device(G.device):
        stream = torch.cuda.current_stream().cuda_stream
        rc = L.bkcore_layer_backward_ref(G.data_ptr(), gf.data_ptr() if gf is not None else None,
                                         gg.data_ptr() if gg is not None else None, float(alpha),
                                         float(feature_clamp) if feature_clamp is not None else 0.0, v.data_ptr(),
                                         float(v_max), 1 if round_bf16 else 0, n, out.data_ptr(), stream)
    _lib.check(rc, "bkcore_layer_backward_ref")
    return out


def sanitise_inputs(he_diag, h0_super, h0_sub):
    """float32 copies of the operands after the operator's input clamp and conditional NaN replacement
    (reference bk_core.py:256-264), plus the masks of elements the clamp left untouched (where the gradient
    passes).  Device-side only: no host synchronisation."""
    a = he_diag.float()
    b = h0_super.float()
    c = h0_sub.float()
    pa = (a.abs() <= 100.0)
    pb = (b.abs() <= 10.0)
    pc = (c.abs() <= 10.0)
    bad = torch.isnan(a).any()
    a = torch.clamp(a, -100.0, 100.0)
    b = torch.clamp(b, -10.0, 10.0)
    c = torch.clamp(c, -10.0, 10.0)
    a = torch.where(bad, torch.nan_to_num(a, nan=0.0), a)
    b = torch.where(bad, torch.nan_to_num(b, nan=1.0), b)
    c = torch.where(bad, torch.nan_to_num(c, nan=1.0), c)
    return a, b, c, pa, pb, pc


def bk_backward_analytic(a, b, c, z, grad_features, grad_G, phi_exact: bool = False, want_bc: bool = True, G=None,
                         states=None):
    """Launch the analytic backward.  a (B,N), b/c (B,N-1): float32, already sanitised.  G: the (B,N) complex64 the
    forward returned for these operands (None: it is computed again inside the call).  states: the boundary states
    that forward call kept (bk_forward(..., want_states=True)); the ratio chains then start from them instead of
    being warmed up again.
    Returns (grad_a, grad_b|None, grad_c|None), float32."""
    L = _lib.load()
    if a.dtype != torch.float32 or b.dtype != torch.float32 or c.dtype != torch.float32:
        raise TypeError("bkcore: bk_backward_analytic takes float32 operands")
    if not a.is_cuda:
        raise RuntimeError("bkcore: tensors must be on a ROCm device (no CPU path)")
    B, N = a.shape
    nb = max(N - 1, 0)
    ga = torch.empty((B, N), dtype=torch.float32, device=a.device)
    gb = torch.empty((B, nb), dtype=torch.float32, device=a.device) if want_bc else None
    gc = torch.empty((B, nb), dtype=torch.float32, device=a.device) if want_bc else None
    if B == 0 or N == 0:
        return ga, gb, gc
    _check_batch(B, b, c)
    a, pa, lda = _rows(a, N, "a")
    b, pb, ldb = _rows(b, nb, "b")
    c, pc, ldc = _rows(c, nb, "c")
    gf = None
    if grad_features is not None:
        gf = grad_features.float().contiguous()
        if tuple(gf.shape) != (B, N, 2):
            raise ValueError("bkcore: grad_features must have shape (B, N, 2)")
    gg = None
    if grad_G is not None:
        gg = grad_G.to(torch.complex64).contiguous()
        if tuple(gg.shape) != (B, N):
            raise ValueError("bkcore: grad_G must have shape (B, N)")
    gsaved = None
    if G is not None:
        if G.dtype != torch.complex64 or tuple(G.shape) != (B, N) or G.device != a.device:
            raise ValueError("bkcore: G must be the (B, N) complex64 output of the forward, on the operands' device")
        gsaved = G.contiguous()
    if states is not None:
        if gsaved is None:
            raise ValueError("bkcore: boundary states come with the G of the same forward call")
        want = (B, int(L.bkcore_boundary_state_bytes(B, N)) // (16 * B), 4)
        if states.dtype != torch.float32 or tuple(states.shape) != want or states.device != a.device \
                or not states.is_contiguous():
            raise ValueError(f"bkcore: states must be the {want} float32 tensor the forward returned")
    with _on_device(a.device):
        stream = torch.cuda.current_stream().cuda_stream
        zkeep, pz, zre, zim = _prep_z(z, a.device)
        state = _state_block(a.device, stream)
        ws_bytes = _ws_ana_bytes.get((B, N))
        if ws_bytes is None:
            ws_bytes = int(L.bkcore_backward_analytic_workspace_bytes(B, N))
            _ws_ana_bytes[(B, N)] = ws_bytes
        if _DEBUG["keep_ws"]:       # diagnostics read the workspace after the call: give this call its own
            ws = torch.empty(max(ws_bytes, 256), dtype=torch.uint8, device=a.device)
        else:
            ws = _cached_workspace(_ws_ana, a.device, stream, ws_bytes)
        rc = L.bkcore_backward_analytic_states(pa, lda, pb if nb else None, ldb, pc if nb else None, ldc, pz, zre, zim, B, N,
                                        gsaved.data_ptr() if gsaved is not None else None,
                                        states.data_ptr() if states is not None else None,
                                        states.numel() * 4 if states is not None else 0,
                                        gf.data_ptr() if gf is not None else None,
                                        gg.data_ptr() if gg is not None else None, ga.data_ptr(),
                                        gb.data_ptr() if gb is not None and nb else None,
                                        gc.data_ptr() if gc is not None and nb else None, ws.data_ptr(), ws_bytes,
                                        state.data_ptr(), _lib.FLAG_PHI_EXACT if phi_exact else 0, stream)
    _lib.check(rc, "bkcore_backward_analytic")
    if _DEBUG["keep_ws"]:
        _DEBUG["last_ana_ws"], _DEBUG["last_ana_shape"] = ws, (B, N)
    del zkeep
    return ga, gb, gc


def analytic_rows_recomputed() -> int:
    """Diagnostics: number of rows of the last analytic backward (run with _DEBUG['keep_ws'] = True) whose in-kernel
    checks failed, so that the three-pass path recomputed them."""
    ws = _DEBUG.get("last_ana_ws")
    if ws is None:
        raise RuntimeError("bkcore: no analytic backward was run with _DEBUG['keep_ws'] = True")
    B, N = _DEBUG["last_ana_shape"]
    off = int(_lib.load().bkcore_backward_analytic_rowflag_offset(B, N))
    return int(ws[off: off + 4 * B].view(torch.int32).ne(0).sum().item())


_ws_complex = {}              # (device index, stream) -> workspace of the complex-operand kernel


def _get_diag_complex(a, b, c, z, phi_exact: bool):
    """The raw resolvent diagonal for complex operands: any of a, b, c complex (the rest is converted, as the reference
    converts every operand to complex128, bk_core.py:45-47; the kernel reads complex64)."""
    L = _lib.load()
    if not a.is_cuda:
        raise RuntimeError("bkcore: tensors must be on a ROCm device (no CPU path; use the oracle for CPU checks)")
    if b.device != a.device or c.device != a.device:
        raise RuntimeError("bkcore: a, b, c must be on the same device")
    a, b, c = (x.detach().to(torch.complex64).resolve_conj().resolve_neg() for x in (a, b, c))
    B, N = a.shape
    nb = max(N - 1, 0)
    G = torch.empty((B, N), dtype=torch.complex64, device=a.device)
    if B == 0 or N == 0:
        return G
    _check_batch(B, b, c)
    a, pa, lda = _rows(a, N, "a")
    b, pb, ldb = _rows(b, nb, "b")
    c, pc, ldc = _rows(c, nb, "c")
    dev = a.device
    with _on_device(dev):
        stream = torch.cuda.current_stream().cuda_stream
        zkeep, pz, zre, zim = _prep_z(z, dev)
        need = int(L.bkcore_get_diag_complex_workspace_bytes(B, N))
        ws = _cached_workspace(_ws_complex, dev, stream, need)
        rc = L.bkcore_get_diag_complex(pa, lda, pb if nb else None, ldb, pc if nb else None, ldc, pz, zre, zim, B, N,
                                       G.data_ptr(), ws.data_ptr(), need, _lib.FLAG_PHI_EXACT if phi_exact else 0,
                                       stream)
    _lib.check(rc, "bkcore_get_diag_complex")
    del zkeep
    return G


def vmapped_get_diag(a, b, c, z):
    """Batched diag((T - zI)^-1): (B,N),(B,N-1),(B,N-1),z -> (B,N) complex64.  No input clamps
    (reference bk_core.py:214-216).  Complex a, b or c are accepted as in the reference (bk_core.py:45-47)."""
    if a.is_complex() or b.is_complex() or c.is_complex():
        return _get_diag_complex(a, b, c, z, BKCoreFunction.PHI_K_INDEX == "exact")
    return bk_forward(a, b, c, z, raw=True, phi_exact=(BKCoreFunction.PHI_K_INDEX == "exact"),
                      want_features=False)[1]


def get_tridiagonal_inverse_diagonal(a, b, c, z):
    """One row: a (N,), b (N-1,), c (N-1,) -> (N,) complex64 (reference bk_core.py:27-210)."""
    return vmapped_get_diag(a.unsqueeze(0), b.unsqueeze(0), c.unsqueeze(0), z)[0]


class BKCoreFunction(torch.autograd.Function):
    """BK-Core autograd operator with the reference's signature (bk_core.py:219-385).

    Class attributes kept from the reference: ``GRAD_BLEND`` (read at backward time, bk_core.py:366),
    ``USE_TRITON`` and ``USE_PARALLEL_SCAN`` (accepted, select nothing).
    Added here: ``PHI_K_INDEX`` in {"reference", "exact"} chooses the off-diagonal product used by the
    phi recurrence (the reference uses k_i, bk_core.py:171).  ``GRAD_MODE`` in {"reference", "analytic"}, read at
    forward time: "reference" is the blend of bk_core.py:303-385 (gradient for he_diag only); "analytic" returns
    the gradient of the resolvent diagonal itself for he_diag, h0_super and h0_sub.
    """
    GRAD_BLEND = 0.5
    USE_TRITON = None
    USE_PARALLEL_SCAN = True
    PHI_K_INDEX = "reference"
    GRAD_MODE = "reference"
    WARN_NONFINITE_GRAD = False     # True: check the incoming gradient on the host (one synchronisation per backward)
                                    # and warn as bk_core.py:321-325 does; "lazy": the kernel raises a device-side
                                    # flag, read back asynchronously -- the same warning, emitted by a later backward
                                    # or by poll_nonfinite_grad_warning(), with no synchronisation
    ADAPTIVE_PATH = False           # True: a call whose predecessor (same device and stream) had to recompute more than
                                    # half of its rows with the scan kernel skips the fast path's arithmetic; every
                                    # 16th such call tries it again.  Which kernel computes a well-conditioned row then
                                    # depends on the calls before it (same numbers to the operator's tolerance, not
                                    # bit for bit), hence off by default
    ROW_UNITS = "auto"              # "auto" | "never" | "always": rows in two units of one launch when there are more
                                    # rows than the GPU holds workgroups at once ("auto": not inside a distributed job,
                                    # see _whole_rows_only)
    KEEP_BOUNDARY_STATES = True     # GRAD_MODE "analytic": the forward also keeps the values its ratio chains carry
                                    # into every block of 8 elements (2 bytes per element, held until the backward)
                                    # and the backward starts its chains from them instead of warming them up again.
                                    # Measured at B = 1024, N = 32768: forward 0.150 -> 0.165 ms, backward 0.361 ->
                                    # 0.325 ms, together 0.511 -> 0.490 ms.  False: nothing but G is kept
    SHARE_FEATURES_STORAGE = True   # features = the float32 view of G_ii (same numbers, bk_core.py:296-298) instead of
                                    # a second tensor: halves the bytes the forward writes.  An in-place modification
                                    # of either output then raises in autograd; False gives two separate tensors

    @staticmethod
    def forward(ctx, he_diag, h0_super, h0_sub, z, use_triton=None, use_parallel_scan=None):
        lead = he_diag.shape[:-1]
        n = he_diag.shape[-1]
        a2 = he_diag.reshape(-1, n) if he_diag.dim() != 2 else he_diag
        b2 = h0_super.reshape(-1, max(n - 1, 0)) if h0_super.dim() != 2 else h0_super
        c2 = h0_sub.reshape(-1, max(n - 1, 0)) if h0_sub.dim() != 2 else h0_sub
        keep = (BKCoreFunction.GRAD_MODE == "analytic" and BKCoreFunction.KEEP_BOUNDARY_STATES
                and any(ctx.needs_input_grad[:3]))
        out = bk_forward(a2.detach(), b2.detach(), c2.detach(), z, raw=False,
                         phi_exact=(BKCoreFunction.PHI_K_INDEX == "exact"),
                         share_storage=BKCoreFunction.SHARE_FEATURES_STORAGE, want_states=keep)
        feats, G = out[0], out[1]
        states = out[2] if keep else None
        if he_diag.dim() != 2:
            feats = feats.reshape(*lead, n, 2)
            G = G.reshape(*lead, n)
        ctx.set_materialize_grads(False)               # an unused output arrives as None, not as a zero tensor
        ctx.bk_mode = BKCoreFunction.GRAD_MODE
        if ctx.bk_mode == "analytic":
            ctx.bk_phi_exact = BKCoreFunction.PHI_K_INDEX == "exact"
            ctx.bk_z = z.detach() if torch.is_tensor(z) else z
            ctx.bk_shapes = (he_diag.shape, h0_super.shape, h0_sub.shape)
            # besides G: the values the ratio chains of every block of 8 elements start from (2 bytes per element), when
            # the forward's path keeps them; the backward recomputes the ratios from the operands, from there
            ctx.bk_states = states
            ctx.save_for_backward(he_diag, h0_super, h0_sub, G)
        elif ctx.bk_mode == "reference":
            ctx.save_for_backward(G)                   # only G is saved (bk_core.py:293)
        else:
            raise ValueError(f"bkcore: unknown GRAD_MODE {ctx.bk_mode!r}")
        return feats, G

    @staticmethod
    def backward(ctx, grad_output_features, grad_G_ii):
        if grad_output_features is None and grad_G_ii is None:
            return None, None, None, None, None, None
        # NaN / Inf in the incoming gradient count as 0 / +-max inside the kernels (torch.nan_to_num semantics,
        # bk_core.py:325); the reference also warns, which costs a host synchronisation, so that is opt-in
        lazy = BKCoreFunction.WARN_NONFINITE_GRAD == "lazy"
        if BKCoreFunction.WARN_NONFINITE_GRAD and not lazy and grad_output_features is not None \
                and not torch.isfinite(grad_output_features).all():
            warnings.warn(_NONFINITE_MSG, RuntimeWarning)
            logger.warning(_NONFINITE_MSG)
        if ctx.bk_mode == "analytic":
            a, b, c, G = ctx.saved_tensors
            n = a.shape[-1]
            nb = max(n - 1, 0)
            rows = a.numel() // max(n, 1)
            a2, b2, c2, pa, pb, pc = sanitise_inputs(a.reshape(rows, n), b.reshape(rows, nb), c.reshape(rows, nb))
            gf = grad_output_features.reshape(-1, n, 2) if grad_output_features is not None else None
            gg = grad_G_ii.reshape(-1, n) if grad_G_ii is not None else None
            want_bc = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
            ga, gb, gc = bk_backward_analytic(a2, b2, c2, ctx.bk_z, gf, gg,
                                              phi_exact=ctx.bk_phi_exact, want_bc=want_bc, G=G.reshape(rows, n),
                                              states=ctx.bk_states)
            sa, sb, sc = ctx.bk_shapes
            ga = (ga * pa).reshape(sa) if ctx.needs_input_grad[0] else None
            gb_o = (gb * pb).reshape(sb) if ctx.needs_input_grad[1] else None
            gc_o = (gc * pc).reshape(sc) if ctx.needs_input_grad[2] else None
            return ga, gb_o, gc_o, None, None, None
        (G,) = ctx.saved_tensors
        grad = bk_backward_reference(G, grad_output_features, grad_G_ii, BKCoreFunction.GRAD_BLEND,
                                     flag_nonfinite=lazy)
        return grad, None, None, None, None, None


class FusedBKLayerFunction(torch.autograd.Function):
    """The BK-Core branch of the reference layer from the output of ``v_proj`` to the clamped features
    (src/models/resnet_bk.py:139-170, 216-217) with its elementwise steps fused into two launches around the operator:

        apply(v_pre (B,N) f32, h0_diag (N), h0_super, h0_sub (B|1, N-1), z, v_max, feature_clamp|None)
            -> (clamp(features) (B,N,2) f32, G_ii (B,N) complex64)

    Forward: one prologue launch (nan_to_num, tanh saturation, + h0_diag, bfloat16), then the operator on bfloat16
    operands with the clamp folded into its store of the features.  Backward: one launch for the whole chain (reference-mode gradient only; the
    gradient reaches ``v_pre`` alone, as in the reference where the off-diagonals are buffers).  ``GRAD_BLEND`` and
    ``PHI_K_INDEX`` are read from ``BKCoreFunction``."""

    @staticmethod
    def forward(ctx, v_pre, h0_diag, h0_super, h0_sub, z, v_max, feature_clamp):
        if BKCoreFunction.GRAD_MODE != "reference":
            raise RuntimeError("bkcore: FusedBKLayerFunction implements the reference-mode gradient only")
        B, N = v_pre.shape
        v = v_pre.detach().float().contiguous()
        he = layer_prologue(v, h0_diag.detach(), v_max, torch.bfloat16)
        nb = max(N - 1, 0)
        if nb == 0:
            b = c = torch.empty((B, 0), dtype=torch.bfloat16, device=v.device)
        else:
            b = h0_super.detach().to(torch.bfloat16).reshape(-1, nb)
            c = h0_sub.detach().to(torch.bfloat16).reshape(-1, nb)
        if b.shape[0] != B:
            b = b.expand(B, -1)
        if c.shape[0] != B:
            c = c.expand(B, -1)
        # without a clamp the features are the float32 view of G: nothing more for the operator to write
        feats, G = bk_forward(he, b, c, z, raw=False, phi_exact=(BKCoreFunction.PHI_K_INDEX == "exact"),
                              feature_clamp=feature_clamp, share_storage=feature_clamp is None)
        ctx.set_materialize_grads(False)
        ctx.bk_v_max = float(v_max)
        ctx.bk_clamp = feature_clamp
        ctx.save_for_backward(G, v)
        return feats, G

    @staticmethod
    def backward(ctx, grad_features, grad_G_ii):
        if grad_features is None and grad_G_ii is None:
            return (None,) * 7
        G, v = ctx.saved_tensors
        grad_v = layer_backward_reference(G, grad_features, grad_G_ii, BKCoreFunction.GRAD_BLEND, ctx.bk_clamp, v,
                                          ctx.bk_v_max, round_bf16=True)
        return grad_v, None, None, None, None, None, None


class FusedSpectralOutputFunction(torch.autograd.Function):
    """out = residual + bk_scale * output_proj(features) of the reference layer (src/models/resnet_bk.py:218-219) in
    one launch forward and one pass over the incoming gradient backward.

        apply(features (..., 2) f32, weight (D, 2), bias (D,), bk_scale (D,), residual (..., D) f32 | None
              [, feature_clamp]) -> (..., D) f32

    feature_clamp: the layer's ``torch.clamp(features, -c, c)`` (resnet_bk.py:216-217) is applied as the kernels read the
    features, which then may be the operator's unclamped output (the float32 view of G_ii): no clamped copy is written,
    and the gradient returned for ``features`` is the one with respect to the unclamped values."""

    @staticmethod
    def forward(ctx, features, weight, bias, scale, residual, feature_clamp=None):
        L = _lib.load()
        if not features.is_cuda:
            raise RuntimeError("bkcore: tensors must be on a ROCm device (no CPU path)")
        D = weight.shape[0]
        if weight.shape != (D, 2) or bias.shape != (D,) or scale.shape != (D,) or features.shape[-1] != 2:
            raise ValueError("bkcore: expected features (..., 2), weight (D, 2), bias (D,), bk_scale (D,)")
        if D > 1024:
            raise ValueError("bkcore: FusedSpectralOutputFunction supports D <= 1024")
        lead = features.shape[:-1]
        f = features.detach().float().contiguous()
        w, bb, sc = (t.detach().float().contiguous() for t in (weight, bias, scale))
        n = f.numel() // 2
        res = None
        if residual is not None:
            if residual.shape != (*lead, D):
                raise ValueError(f"bkcore: residual must have shape {(*lead, D)}, got {tuple(residual.shape)}")
            res = residual.detach().float().contiguous()
        out = torch.empty((*lead, D), dtype=torch.float32, device=f.device)
        fc = 0.0
        if feature_clamp is not None:
            fc = float(feature_clamp)
            if not fc > 0.0:
                raise ValueError("bkcore: feature_clamp must be positive")
        if n and D:
            with _on_device(f.device):
                rc = L.bkcore_layer_epilogue_clamped(f.data_ptr(), fc, w.data_ptr(), bb.data_ptr(), sc.data_ptr(),
                                                     res.data_ptr() if res is not None else None, n, D,
                                                     out.data_ptr(), torch.cuda.current_stream().cuda_stream)
            _lib.check(rc, "bkcore_layer_epilogue")
        ctx.save_for_backward(f, w, bb, sc)
        ctx.bk_has_res = residual is not None
        ctx.bk_fc = fc
        return out

    @staticmethod
    def backward(ctx, grad_out):
        L = _lib.load()
        f, w, bb, sc = ctx.saved_tensors
        D = w.shape[0]
        n = f.numel() // 2
        go = grad_out.float().contiguous()
        gf = torch.empty_like(f)
        need_w, need_b, need_s = ctx.needs_input_grad[1], ctx.needs_input_grad[2], ctx.needs_input_grad[3]
        gw = torch.empty_like(w) if need_w else None
        gb = torch.empty_like(bb) if need_b else None
        gs = torch.empty_like(sc) if need_s else None
        if n and D:
            with _on_device(f.device):
                nbytes = int(L.bkcore_layer_epilogue_backward_workspace_bytes(n, D))
                ws = torch.empty(nbytes, dtype=torch.uint8, device=f.device)
                ptr = lambda t: t.data_ptr() if t is not None else None
                rc = L.bkcore_layer_epilogue_backward_clamped(go.data_ptr(), f.data_ptr(), ctx.bk_fc, w.data_ptr(),
                                                              bb.data_ptr(), sc.data_ptr(), n, D, gf.data_ptr(),
                                                              ptr(gw), ptr(gb), ptr(gs), ws.data_ptr(), nbytes,
                                                              torch.cuda.current_stream().cuda_stream)
            _lib.check(rc, "bkcore_layer_epilogue_backward")
        else:
            for t in (gf, gw, gb, gs):
                if t is not None:
                    t.zero_()
        return (gf if ctx.needs_input_grad[0] else None, gw, gb, gs, go if ctx.bk_has_res else None, None)


def set_triton_mode(enabled: bool):
    """Kept for source compatibility (reference bk_core.py:389-416).  There is no Triton back-end here;
    the flag is recorded and selects nothing."""
    BKCoreFunction.USE_TRITON = bool(enabled)
    if enabled:
        warnings.warn("Triton is not used by this build; BK-Core always runs the HIP kernels.", UserWarning)
        BKCoreFunction.USE_TRITON = False


def get_triton_mode() -> bool:
    """Always False: no Triton back-end (reference bk_core.py:419-433)."""
    return False
