"""Cumulative sum along one dimension on the GPU: the causal accumulation of the reference's AR-SSM layer
(src/models/phase1/ar_ssm_layer.py:517-525) behind the signature of its ``fused_associative_scan``
(src/kernels/associative_scan.py:211-308).  Differentiable: the gradient of a cumulative sum is the cumulative sum of
the incoming gradient in the opposite direction."""
from __future__ import annotations

import math

import torch

from . import _lib


def _cumsum_launch(x: torch.Tensor, dim: int, reverse: bool) -> torch.Tensor:
    L = _lib.load()
    if not x.is_cuda:
        raise RuntimeError("bkcore: tensors must be on a ROCm device (no CPU path)")
    if x.is_complex() or not x.is_floating_point():
        raise TypeError("bkcore: fused_associative_scan takes real floating-point tensors")
    src = x if x.dtype == torch.float32 else x.float()
    src = src.contiguous()
    out = torch.empty_like(src)
    if src.numel():
        shape = src.shape
        outer = math.prod(shape[:dim])
        inner = math.prod(shape[dim + 1:])
        with torch.cuda.device(src.device):
            rc = L.bkcore_cumsum(src.data_ptr(), out.data_ptr(), outer, shape[dim], inner, 1 if reverse else 0,
                                 torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "bkcore_cumsum")
    return out if x.dtype == torch.float32 else out.to(x.dtype)


class _CumsumFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dim, reverse):
        ctx.bk_dim, ctx.bk_reverse = dim, reverse
        return _cumsum_launch(x.detach(), dim, reverse)

    @staticmethod
    def backward(ctx, grad_out):
        return _cumsum_launch(grad_out, ctx.bk_dim, not ctx.bk_reverse), None, None


def fused_associative_scan(x: torch.Tensor, dim: int = -1, reverse: bool = False) -> torch.Tensor:
    """Cumulative sum of ``x`` along ``dim`` (``reverse``: from the end), any shape, result in x's dtype (summed in
    float32)."""
    if x.dim() == 0:
        raise ValueError("bkcore: fused_associative_scan needs at least one dimension")
    if not -x.dim() <= dim < x.dim():
        raise IndexError(f"bkcore: dim {dim} out of range for a {x.dim()}-dimensional tensor")
    return _CumsumFunction.apply(x, dim % x.dim(), bool(reverse))
