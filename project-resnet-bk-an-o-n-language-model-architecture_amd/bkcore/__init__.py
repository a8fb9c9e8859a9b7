"""MI355X-native BK-Core: host-side mirror of the reference operator API (src/models/bk_core.py)."""
from .bk_core import (BKCoreFunction, get_tridiagonal_inverse_diagonal, vmapped_get_diag, set_triton_mode,
                      get_triton_mode, bk_forward, bk_backward_reference, bk_backward_analytic,
                      sanitise_inputs, FusedBKLayerFunction, FusedSpectralOutputFunction, layer_prologue, layer_backward_reference,
                      poll_nonfinite_grad_warning)
from .dropin import make_dispatching
from .variants import (BatchedAnalyticBKCoreFunction, MemoryOptimizedBKCoreFunction, BKScanStableFunction,
                       bk_backward_variant)

from .scan import fused_associative_scan
from .birman_schwinger import woodbury_correct, compute_semiseparable_resolvent
from .layers import BKCoreLayer, CUDAOptimizedBKCore, LearnedTridiagonalGreen

__all__ = ["fused_associative_scan", "woodbury_correct", "compute_semiseparable_resolvent", "BKCoreLayer", "CUDAOptimizedBKCore", "LearnedTridiagonalGreen", "BKCoreFunction", "get_tridiagonal_inverse_diagonal", "vmapped_get_diag", "set_triton_mode",
           "get_triton_mode", "bk_forward", "bk_backward_reference", "bk_backward_analytic", "sanitise_inputs", "FusedBKLayerFunction", "FusedSpectralOutputFunction",
           "layer_prologue", "layer_backward_reference", "poll_nonfinite_grad_warning", "make_dispatching", "BatchedAnalyticBKCoreFunction",
           "MemoryOptimizedBKCoreFunction", "BKScanStableFunction", "bk_backward_variant"]
