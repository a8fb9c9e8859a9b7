"""ctypes binding of the C oracle (oracle/bk_oracle.c).  Test infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libbkoracle.so")


def build(force=False):
    src = os.path.join(_HERE, "bk_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libbkoracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        L = ctypes.CDLL(_LIB)
        dp, fp, l, d, i = (ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_float), ctypes.c_long,
                           ctypes.c_double, ctypes.c_int)
        L.bk_oracle_get_diag.argtypes = [dp, l, dp, l, dp, l, d, d, l, l, i, fp, dp]
        L.bk_oracle_get_diag.restype = i
        L.bk_oracle_forward_f32.argtypes = [fp, l, fp, l, fp, l, d, d, l, l, i, fp, fp]
        L.bk_oracle_forward_f32.restype = i
        L.bk_oracle_backward_ref.argtypes = [fp, fp, fp, d, l, fp]
        L.bk_oracle_backward_ref.restype = i
        L.bk_oracle_num_threads.restype = i
        _lib = L
    return _lib


def _p(arr, ct):
    return arr.ctypes.data_as(ctypes.POINTER(ct)) if arr is not None else None


def get_diag(a, b, c, z, phi_exact=False, want128=False):
    """a (B,N), b,c (B,N-1) -> complex64 (B,N) [and complex128 if want128]; no input clamps."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    c = np.ascontiguousarray(c, dtype=np.float64)
    B, N = a.shape
    nb = max(N - 1, 0)
    G = np.zeros((B, N), dtype=np.complex64)
    G128 = np.zeros((B, N), dtype=np.complex128) if want128 else None
    z = complex(z)
    rc = lib().bk_oracle_get_diag(_p(a, ctypes.c_double), N, _p(b, ctypes.c_double), nb, _p(c, ctypes.c_double), nb,
                                  z.real, z.imag, B, N, int(phi_exact),
                                  _p(G.view(np.float32), ctypes.c_float),
                                  _p(G128.view(np.float64), ctypes.c_double) if want128 else None)
    if rc != 0:
        raise RuntimeError(f"bk_oracle_get_diag failed: {rc}")
    return (G, G128) if want128 else G


def forward_f32(a, b, c, z, phi_exact=False):
    """BKCoreFunction.forward semantics (clamp/sanitise) on float32 inputs -> (features, G)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    c = np.ascontiguousarray(c, dtype=np.float32)
    B, N = a.shape
    nb = max(N - 1, 0)
    G = np.zeros((B, N), dtype=np.complex64)
    F = np.zeros((B, N, 2), dtype=np.float32)
    z = complex(z)
    rc = lib().bk_oracle_forward_f32(_p(a, ctypes.c_float), N, _p(b, ctypes.c_float), nb, _p(c, ctypes.c_float), nb,
                                     z.real, z.imag, B, N, int(phi_exact),
                                     _p(G.view(np.float32), ctypes.c_float), _p(F, ctypes.c_float))
    if rc != 0:
        raise RuntimeError(f"bk_oracle_forward_f32 failed: {rc}")
    return F, G


def backward_ref(G, gfeat, gG, alpha):
    G = np.ascontiguousarray(G, dtype=np.complex64)
    n = G.size
    gf = np.ascontiguousarray(gfeat, dtype=np.float32) if gfeat is not None else None
    gg = np.ascontiguousarray(gG, dtype=np.complex64) if gG is not None else None
    out = np.zeros(G.shape, dtype=np.float32)
    rc = lib().bk_oracle_backward_ref(_p(G.view(np.float32), ctypes.c_float), _p(gf, ctypes.c_float),
                                      _p(gg.view(np.float32), ctypes.c_float) if gg is not None else None,
                                      float(alpha), n, _p(out, ctypes.c_float))
    if rc != 0:
        raise RuntimeError(f"bk_oracle_backward_ref failed: {rc}")
    return out


def num_threads():
    return int(lib().bk_oracle_num_threads())
