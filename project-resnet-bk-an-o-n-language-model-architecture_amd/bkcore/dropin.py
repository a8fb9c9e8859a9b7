"""Dispatching drop-in for the reference module ``src/models/bk_core.py``: tensors on a ROCm device go to the HIP
operator of this package, everything else (CPU tensors) to the reference's own PyTorch definitions, which stay in the
module under private names.  What a maintainer adds at the bottom of the reference module::

    from bkcore.dropin import make_dispatching
    BKCoreFunction, vmapped_get_diag, get_tridiagonal_inverse_diagonal = make_dispatching(
        BKCoreFunction, vmapped_get_diag, get_tridiagonal_inverse_diagonal)

Call sites keep ``BKCoreFunction.apply(he_diag, h0_super, h0_sub, z[, use_triton[, use_parallel_scan]])`` and the
class attributes ``GRAD_BLEND`` / ``USE_TRITON`` / ``USE_PARALLEL_SCAN`` (reference bk_core.py:219-246, 366): reads
come from the reference class, writes go to both implementations.
"""
from __future__ import annotations

import torch

_FORWARDED = ("GRAD_BLEND", "USE_TRITON", "USE_PARALLEL_SCAN")


def _on_gpu(x) -> bool:
    return torch.is_tensor(x) and x.is_cuda


def make_dispatching(ref_function, ref_vmapped_get_diag=None, ref_get_diag=None, hip_module=None):
    """Returns (BKCoreFunction, vmapped_get_diag, get_tridiagonal_inverse_diagonal) that route by the device of the
    first tensor argument.  ``ref_function`` is the reference's autograd.Function class (anything with ``apply`` and
    the forwarded class attributes); ``hip_module`` defaults to this package, imported on the first GPU call so that
    a machine without the library still runs the CPU path."""
    holder = {"hip": hip_module}

    def hip():
        if holder["hip"] is None:
            import bkcore
            holder["hip"] = bkcore
        return holder["hip"]

    class _Meta(type):
        def __getattr__(cls, name):                     # anything not defined on the shim: the reference class
            return getattr(ref_function, name)

        def __setattr__(cls, name, value):
            if name in _FORWARDED:
                setattr(ref_function, name, value)
                try:
                    setattr(hip().BKCoreFunction, name, value)
                except (ImportError, RuntimeError, OSError):
                    pass                                # no HIP library here: the CPU path is all there is
            else:
                setattr(ref_function, name, value)

    class BKCoreFunction(metaclass=_Meta):
        """``apply`` with the reference's signature (bk_core.py:219-246)."""

        @staticmethod
        def apply(he_diag, h0_super, h0_sub, z, *flags):
            if _on_gpu(he_diag):
                return hip().BKCoreFunction.apply(he_diag, h0_super, h0_sub, z, *flags)
            return ref_function.apply(he_diag, h0_super, h0_sub, z, *flags)

    def vmapped_get_diag(a, b, c, z):
        if _on_gpu(a):
            return hip().vmapped_get_diag(a, b, c, z)
        return ref_vmapped_get_diag(a, b, c, z)

    def get_tridiagonal_inverse_diagonal(a, b, c, z):
        if _on_gpu(a):
            return hip().get_tridiagonal_inverse_diagonal(a, b, c, z)
        return ref_get_diag(a, b, c, z)

    return BKCoreFunction, vmapped_get_diag, get_tridiagonal_inverse_diagonal
