"""The reference's other autograd wrappers around the same resolvent diagonal, on the HIP kernels:

* ``BatchedAnalyticBKCoreFunction`` / ``MemoryOptimizedBKCoreFunction`` (src/models/batched_gradient.py:65-130,174-217):
  ``apply(he_diag, h0_super, h0_sub, z) -> features (B, N, 2)``; forward = ``vmapped_get_diag`` (no input clamps,
  :91,188); backward = ``compute_single_gradient`` (:12-54) for he_diag, ``None`` for the rest.  The reference's two
  classes differ only in how the batch is cut up on the host (vmap over rows / chunks of 32 rows); one launch here.
* ``BKScanStableFunction`` (src/kernels/bk_scan_stable.py:556-611): ``apply(a, b, c, z_real, z_imag) -> G`` complex64;
  backward ``grad_a = -Re(g) |G|^2`` (:597), zeros for b and c (:605-609): that backward is what is reproduced.
  The reference's forward there (log-polar arithmetic, :39-314) is NOT the operator's recurrence: it keeps
  ``phi_raw[:, 1:].flip(1)`` (:284), i.e. the trailing minor of size N - i where the resolvent diagonal needs size
  N - 1 - i (its G_00 comes out as 1), and it has no diagonal regularisation.  Its numbers are neither the oracle's nor
  the resolvent's and are not matched; the forward here is the operator's own in get-diag mode
  (src/models/bk_core.py:27-210 semantics), whose scaled arithmetic is overflow-proof as well.
"""
from __future__ import annotations

import torch

from . import _lib
from .bk_core import _on_device, bk_forward


def bk_backward_variant(G, grad_features, grad_G, alpha: float, variant: int):
    """Launch the elementwise backward in one of the reference's variants (_lib.BWD_*).  float32 (B, N)."""
    L = _lib.load()
    if not G.is_cuda:
        raise RuntimeError("bkcore: tensors must be on a ROCm device (no CPU path)")
    G = G.contiguous()
    out = torch.empty(G.shape, dtype=torch.float32, device=G.device)
    n = G.numel()
    if n == 0:
        return out
    gf = grad_features.float().contiguous() if grad_features is not None else None
    gg = grad_G.to(torch.complex64).contiguous() if grad_G is not None else None
    with _on_device(G.device):
        rc = L.bkcore_backward_variant(G.data_ptr(), gf.data_ptr() if gf is not None else None,
                                       gg.data_ptr() if gg is not None else None, float(alpha), n, out.data_ptr(),
                                       int(variant), torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "bkcore_backward_variant")
    return out


class BatchedAnalyticBKCoreFunction(torch.autograd.Function):
    GRAD_BLEND = 0.5

    @staticmethod
    def forward(ctx, he_diag, h0_super, h0_sub, z):
        feats, G = bk_forward(he_diag.detach(), h0_super.detach(), h0_sub.detach(), z, raw=True, share_storage=True)
        ctx.save_for_backward(G)
        ctx.bk_cls = BatchedAnalyticBKCoreFunction
        return feats

    @staticmethod
    def backward(ctx, grad_output_features):
        (G,) = ctx.saved_tensors
        grad = bk_backward_variant(G, grad_output_features, None, ctx.bk_cls.GRAD_BLEND, _lib.BWD_BATCHED)
        return grad, None, None, None


class MemoryOptimizedBKCoreFunction(torch.autograd.Function):
    GRAD_BLEND = 0.5
    CHUNK_SIZE = 32      # accepted for compatibility; the whole batch is one launch

    @staticmethod
    def forward(ctx, he_diag, h0_super, h0_sub, z):
        feats, G = bk_forward(he_diag.detach(), h0_super.detach(), h0_sub.detach(), z, raw=True, share_storage=True)
        ctx.save_for_backward(G)
        return feats

    @staticmethod
    def backward(ctx, grad_output_features):
        (G,) = ctx.saved_tensors
        grad = bk_backward_variant(G, grad_output_features, None, MemoryOptimizedBKCoreFunction.GRAD_BLEND,
                                   _lib.BWD_BATCHED)
        return grad, None, None, None


class BKScanStableFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, c, z_real, z_imag):
        z = complex(float(z_real), float(z_imag))
        _, G = bk_forward(a.detach(), b.detach(), c.detach(), z, raw=True, share_storage=True)
        ctx.save_for_backward(G)
        ctx.bk_dtypes = (a.dtype, b, c)
        return G

    @staticmethod
    def backward(ctx, grad_output):
        (G,) = ctx.saved_tensors
        a_dtype, b, c = ctx.bk_dtypes
        grad_a = grad_b = grad_c = None
        if ctx.needs_input_grad[0]:
            grad_a = bk_backward_variant(G, None, grad_output, 0.0, _lib.BWD_SCAN_STABLE).to(a_dtype)
        if ctx.needs_input_grad[1]:
            grad_b = torch.zeros_like(b)
        if ctx.needs_input_grad[2]:
            grad_c = torch.zeros_like(c)
        return grad_a, grad_b, grad_c, None, None
