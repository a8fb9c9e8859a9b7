"""The semiseparable resolvent of the reference's Birman-Schwinger core on the GPU
(src/models/birman_schwinger_core.py:463-608): tridiagonal part by the BK-Core operator (float64 operands as the
reference passes them, :523-540), low-rank part by the batched Woodbury kernels instead of a Python loop over rows."""
from __future__ import annotations

import torch

from . import _lib
from .bk_core import bk_forward


def woodbury_correct(G: torch.Tensor, U: torch.Tensor, V: torch.Tensor) -> torch.Tensor:
    """G (B, N) complex64 on the GPU, U and V (N, r) real: G - G^2 * diag(U (I + V^T diag(G) U)^-1 V^T) per row
    (birman_schwinger_core.py:567-604)."""
    L = _lib.load()
    if not G.is_cuda:
        raise RuntimeError("bkcore: tensors must be on a ROCm device (no CPU path)")
    if G.dim() != 2 or U.dim() != 2 or V.shape != U.shape or U.shape[0] != G.shape[1]:
        raise ValueError("bkcore: expected G (B, N), U and V (N, r)")
    B, N = G.shape
    r = U.shape[1]
    if r > 22:
        raise ValueError("bkcore: woodbury_correct supports rank <= 22")
    g = G.to(torch.complex64).contiguous()
    u = U.detach().to(device=g.device, dtype=torch.float32).contiguous()
    v = V.detach().to(device=g.device, dtype=torch.float32).contiguous()
    out = torch.empty_like(g)
    if B == 0 or N == 0:
        return out
    with torch.cuda.device(g.device):
        nbytes = int(L.bkcore_woodbury_workspace_bytes(B, N, r))
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=g.device)
        rc = L.bkcore_woodbury_correct(g.data_ptr(), u.data_ptr() if r else None, v.data_ptr() if r else None, B, N, r,
                                       out.data_ptr(), ws.data_ptr(), nbytes, torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "bkcore_woodbury_correct")
    return out


def compute_semiseparable_resolvent(v: torch.Tensor, z, U: torch.Tensor | None = None, V: torch.Tensor | None = None,
                                    gamma: float | None = None) -> torch.Tensor:
    """G_ii = diag((H - zI)^-1) for H = tridiag(1, -2 + v, 1) + U V^T, v (B, N) on the GPU -> (B, N) complex64.
    Same steps as BirmanSchwingerCore.compute_semiseparable_resolvent (birman_schwinger_core.py:463-608): float64
    diagonal, z + i gamma, raw get-diag semantics, then the low-rank correction unless U, V are absent or all zero."""
    B, N = v.shape
    he_diag = -2.0 + v.to(torch.float64)
    if gamma is not None:
        z = z + 1j * gamma
    ones = torch.ones(1, max(N - 1, 0), dtype=torch.float64, device=v.device).expand(B, -1)
    G = bk_forward(he_diag, ones, ones, z, raw=True, want_features=False)[1]
    if U is None or V is None or U.shape[1] == 0:
        return G
    return woodbury_correct(G, U, V)
