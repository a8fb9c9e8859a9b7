"""ctypes loader of libbkcore.so (the HIP kernels behind the C ABI declared in include/bkcore.h).

There is no CPU fallback: if the shared library is missing or does not export the expected
symbols, importing the operator fails loudly.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbkcore.so")

F32, BF16, F16, F64 = 0, 1, 2, 3
FLAG_RAW = 1
FLAG_PHI_EXACT = 2
FLAG_SCAN_PATH = 4
FLAG_ADAPTIVE = 8
FLAG_ROW_UNITS = 16
FLAG_ROW_UNITS_MANY = 32
FLAG_ROW_UNITS_MUTE = 64
FLAG_WHOLE_ROWS = 128
FLAG_ROWS_UNCUT = 256      # diagnostics: one workgroup per row whatever the shape
STATE_BYTES = 64
ABI_VERSION = 18
EINVAL, ETOOLONG, ELAUNCH, EWORKSPACE, ETOOBIG = -1, -2, -3, -4, -5
BWD_REFERENCE, BWD_BATCHED, BWD_SCAN_STABLE = 0, 1, 2
NOALLOC_MAX_WORKSPACE = 8 << 20

EXPORTS = ("bkcore_forward", "bkcore_forward_clamped", "bkcore_forward_noalloc", "bkcore_forward_states",
           "bkcore_boundary_state_bytes", "bkcore_backward_analytic_states", "bkcore_workspace_bytes", "bkcore_backward_ref", "bkcore_backward_ref_flagged", "bkcore_backward_variant", "bkcore_backward_analytic",
           "bkcore_backward_analytic_workspace_bytes", "bkcore_backward_analytic_rowflag_offset", "bkcore_layer_prologue", "bkcore_layer_backward_ref",
           "bkcore_layer_epilogue", "bkcore_layer_epilogue_backward", "bkcore_layer_epilogue_backward_workspace_bytes",
           "bkcore_layer_epilogue_clamped", "bkcore_layer_epilogue_backward_clamped", "bkcore_cumsum", "bkcore_woodbury_correct", "bkcore_woodbury_workspace_bytes",
           "bkcore_get_diag_complex", "bkcore_get_diag_complex_workspace_bytes",
           "bkcore_max_seq_len", "bkcore_forward_plan", "bkcore_last_hip_error",
           "bkcore_error_string", "bkcore_abi_version")

_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"bkcore: {LIB_PATH} not found. Build it with `make -C <package>/csrc` (hipcc --offload-arch=gfx950); "
            "there is no CPU fallback for this operator.")
    L = ctypes.CDLL(LIB_PATH)
    vp, i64, i32, f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float
    L.bkcore_forward.argtypes = [vp, i64, vp, i64, vp, i64, i32, vp, f32, f32, i64, i64, vp, vp, vp, ctypes.c_size_t,
                                 vp, ctypes.c_uint, vp]
    L.bkcore_forward_clamped.argtypes = [vp, i64, vp, i64, vp, i64, i32, vp, f32, f32, i64, i64, vp, vp, f32, vp,
                                         ctypes.c_size_t, vp, ctypes.c_uint, vp]
    L.bkcore_forward_clamped.restype = i32
    L.bkcore_forward_noalloc.argtypes = [vp, i64, vp, i64, vp, i64, i32, vp, f32, f32, i64, i64, vp, vp, f32,
                                         ctypes.c_uint, vp]
    L.bkcore_forward_noalloc.restype = i32
    L.bkcore_workspace_bytes.argtypes = [i64, i64]
    L.bkcore_workspace_bytes.restype = ctypes.c_size_t
    L.bkcore_forward.restype = i32
    L.bkcore_backward_ref.argtypes = [vp, vp, vp, f32, i64, vp, vp]
    L.bkcore_backward_ref.restype = i32
    L.bkcore_backward_ref_flagged.argtypes = [vp, vp, vp, f32, i64, vp, vp, vp]
    L.bkcore_backward_ref_flagged.restype = i32
    L.bkcore_backward_variant.argtypes = [vp, vp, vp, f32, i64, vp, i32, vp]
    L.bkcore_backward_variant.restype = i32
    L.bkcore_backward_analytic.argtypes = [vp, i64, vp, i64, vp, i64, vp, f32, f32, i64, i64, vp, vp, vp, vp, vp, vp,
                                           vp, ctypes.c_size_t, vp, ctypes.c_uint, vp]
    L.bkcore_backward_analytic.restype = i32
    L.bkcore_backward_analytic_states.argtypes = [vp, i64, vp, i64, vp, i64, vp, f32, f32, i64, i64, vp, vp,
                                                  ctypes.c_size_t, vp, vp, vp, vp, vp, vp, ctypes.c_size_t, vp,
                                                  ctypes.c_uint, vp]
    L.bkcore_backward_analytic_states.restype = i32
    L.bkcore_forward_states.argtypes = [vp, i64, vp, i64, vp, i64, i32, vp, f32, f32, i64, i64, vp, vp, f32, vp,
                                        ctypes.c_size_t, ctypes.POINTER(ctypes.c_int), vp, ctypes.c_size_t, vp,
                                        ctypes.c_uint, vp]
    L.bkcore_forward_states.restype = i32
    L.bkcore_boundary_state_bytes.argtypes = [i64, i64]
    L.bkcore_boundary_state_bytes.restype = ctypes.c_size_t
    L.bkcore_backward_analytic_workspace_bytes.argtypes = [i64, i64]
    L.bkcore_backward_analytic_workspace_bytes.restype = ctypes.c_size_t
    L.bkcore_backward_analytic_rowflag_offset.argtypes = [i64, i64]
    L.bkcore_backward_analytic_rowflag_offset.restype = ctypes.c_size_t
    L.bkcore_layer_prologue.argtypes = [vp, i64, vp, f32, i64, i64, vp, i32, vp]
    L.bkcore_layer_prologue.restype = i32
    L.bkcore_layer_backward_ref.argtypes = [vp, vp, vp, f32, f32, vp, f32, i32, i64, vp, vp]
    L.bkcore_layer_backward_ref.restype = i32
    L.bkcore_layer_epilogue.argtypes = [vp, vp, vp, vp, vp, i64, i64, vp, vp]
    L.bkcore_layer_epilogue.restype = i32
    L.bkcore_layer_epilogue_backward.argtypes = [vp, vp, vp, vp, vp, i64, i64, vp, vp, vp, vp, vp, ctypes.c_size_t, vp]
    L.bkcore_layer_epilogue_backward.restype = i32
    L.bkcore_layer_epilogue_clamped.argtypes = [vp, f32, vp, vp, vp, vp, i64, i64, vp, vp]
    L.bkcore_layer_epilogue_clamped.restype = i32
    L.bkcore_layer_epilogue_backward_clamped.argtypes = [vp, vp, f32, vp, vp, vp, i64, i64, vp, vp, vp, vp, vp,
                                                         ctypes.c_size_t, vp]
    L.bkcore_layer_epilogue_backward_clamped.restype = i32
    L.bkcore_layer_epilogue_backward_workspace_bytes.argtypes = [i64, i64]
    L.bkcore_layer_epilogue_backward_workspace_bytes.restype = ctypes.c_size_t
    L.bkcore_cumsum.argtypes = [vp, vp, i64, i64, i64, i32, vp]
    L.bkcore_cumsum.restype = i32
    L.bkcore_woodbury_correct.argtypes = [vp, vp, vp, i64, i64, i64, vp, vp, ctypes.c_size_t, vp]
    L.bkcore_woodbury_correct.restype = i32
    L.bkcore_woodbury_workspace_bytes.argtypes = [i64, i64, i64]
    L.bkcore_woodbury_workspace_bytes.restype = ctypes.c_size_t
    L.bkcore_get_diag_complex.argtypes = [vp, i64, vp, i64, vp, i64, vp, f32, f32, i64, i64, vp, vp, ctypes.c_size_t,
                                          ctypes.c_uint, vp]
    L.bkcore_get_diag_complex.restype = i32
    L.bkcore_get_diag_complex_workspace_bytes.argtypes = [i64, i64]
    L.bkcore_get_diag_complex_workspace_bytes.restype = ctypes.c_size_t
    L.bkcore_max_seq_len.restype = i64
    L.bkcore_forward_plan.argtypes = [i64, i64, i32, ctypes.c_uint, i32, ctypes.POINTER(ctypes.c_int)]
    L.bkcore_forward_plan.restype = i32
    L.bkcore_last_hip_error.restype = i32
    L.bkcore_error_string.argtypes = [i32]
    L.bkcore_error_string.restype = ctypes.c_char_p
    L.bkcore_abi_version.restype = i32
    if L.bkcore_abi_version() != ABI_VERSION:
        raise RuntimeError(f"bkcore: ABI version mismatch (library {L.bkcore_abi_version()}, wrapper {ABI_VERSION})")
    _lib = L
    return L


def check(rc, what):
    if rc != 0:
        L = load()
        msg = L.bkcore_error_string(rc).decode()
        raise RuntimeError(f"bkcore: {what} failed: {msg} (code {rc}, hip error {L.bkcore_last_hip_error()})")
