"""CPU oracle for BK-Core (test infrastructure only -- never imported by the product path).

A plain restatement, in batched torch complex128, of the reference's sequential
log-scaled theta/phi recurrences and of the ``BKCoreFunction`` forward/backward
semantics.  It has the same cost model as the reference CPU path (a Python loop
over the sequence with ~20 small tensor ops per step, every op vectorised over
the batch), which is why ``bench.py`` also times it as a CPU baseline.

Reference being restated (paths relative to the reference checkout):
  * src/models/bk_core.py:27-210   get_tridiagonal_inverse_diagonal (one row)
  * src/models/bk_core.py:214-216  vmapped_get_diag (batched)
  * src/models/bk_core.py:239-300  BKCoreFunction.forward (clamp / sanitise / features)
  * src/models/bk_core.py:303-385  BKCoreFunction.backward (blend of -Re(g* G^2) and -Re(g*/G^2))

Parity of this file against the reference is pinned by the fixtures in
``tests/golden`` (made by ``oracle/gen_golden.py``, which imports the reference
in the build container) -- see ``tests/test_oracle_golden.py``.

Everything here is differentiable by torch autograd, so the "analytic" gradient
oracle (true d/da, d/db, d/dc through the regularised, clamped forward) is just
``torch.autograd.grad`` through :func:`get_diag`.
"""
from __future__ import annotations

import torch

WALL_K = 88.0          # bk_core.py:60,198  soft clamp threshold
MASS_EPS = 1e-3        # bk_core.py:109     diagonal regularisation
MAX_MAG = 50.0         # bk_core.py:205     resolvent magnitude clip


def _regularise(x: torch.Tensor) -> torch.Tensor:
    """x + 1e-3 * x/|x| (direction 1 when |x| < 1e-12).  bk_core.py:78-94."""
    mag = x.abs()
    direction = torch.where(mag < 1e-12, torch.ones_like(x), x / (mag + 1e-12))
    return x + MASS_EPS * direction


def _wall(diff: torch.Tensor) -> torch.Tensor:
    """exp(88 * tanh(diff / 88)).  bk_core.py:60-75."""
    return torch.exp(WALL_K * torch.tanh(diff / WALL_K))


def _as_c128(z, device) -> torch.Tensor:
    if isinstance(z, (int, float, complex)):
        return torch.tensor(z, dtype=torch.complex128, device=device)
    return z.to(torch.complex128)


def get_diag(a: torch.Tensor, b: torch.Tensor, c: torch.Tensor, z,
             phi_k_index: str = "reference", out_dtype=torch.complex64) -> torch.Tensor:
    """Batched restatement of bk_core.py:27-216.

    a: (B, N); b, c: (B, N-1); z complex scalar.  Returns (B, N) ``out_dtype``.

    ``phi_k_index="reference"`` uses k_i = c_i b_i when forming phi_i from
    a_{i+1}, phi_{i+1}, phi_{i+2} exactly as bk_core.py:165-171 does;
    ``"exact"`` uses k_{i+1}, which is the true trailing-minor recurrence.
    """
    B, n = a.shape
    dev = a.device
    if n == 0:                                                   # bk_core.py:56-57
        return torch.zeros(B, 0, dtype=out_dtype, device=dev)
    ac = a.to(torch.complex128) - _as_c128(z, dev)               # bk_core.py:45-54
    k = (c.to(torch.complex128) * b.to(torch.complex128))        # bk_core.py:125,171
    ar = _regularise(ac)                                         # every diagonal entry, both sweeps

    one = torch.ones(B, dtype=torch.complex128, device=dev)
    zero_l = torch.zeros(B, dtype=torch.float64, device=dev)

    # theta sweep, bk_core.py:100-140: theta_i = t_i * exp(l_i), |t_i| = 1
    t_th = [one]
    l_th = [zero_l]
    v = ar[:, 0]
    s = v.abs() + 1e-20
    t_th.append(v / s)
    l_th.append(s.log())
    for i in range(1, n):
        v = ar[:, i] * t_th[-1] - k[:, i - 1] * t_th[-2] * _wall(l_th[-2] - l_th[-1])
        s = v.abs() + 1e-20
        l_new = l_th[-1] + s.log()
        t_th.append(v / s)
        l_th.append(l_new)
    t_det, l_det = t_th[-1], l_th[-1]                            # bk_core.py:145-147

    # phi sweep, bk_core.py:150-184
    t_ph = [None] * n
    l_ph = [None] * n
    t_ph[n - 1] = one
    l_ph[n - 1] = zero_l
    if n > 1:
        v = ar[:, n - 1]
        s = v.abs() + 1e-20
        t_ph[n - 2] = v / s
        l_ph[n - 2] = s.log()
        for i in range(n - 3, -1, -1):
            kk = k[:, i] if phi_k_index == "reference" else k[:, i + 1]
            v = ar[:, i + 1] * t_ph[i + 1] - kk * t_ph[i + 2] * _wall(l_ph[i + 2] - l_ph[i + 1])
            s = v.abs() + 1e-20
            t_ph[i] = v / s
            l_ph[i] = l_ph[i + 1] + s.log()

    tt = torch.stack(t_th[:-1], dim=1)
    lt = torch.stack(l_th[:-1], dim=1)
    tp = torch.stack(t_ph, dim=1)
    lp = torch.stack(l_ph, dim=1)

    # combine, bk_core.py:190-201
    logmag = WALL_K * torch.tanh((lt + lp - l_det[:, None]) / WALL_K)
    g = (tt * tp) / (t_det[:, None] + 1e-18) * torch.exp(logmag.to(torch.complex128))
    # bk_core.py:204-208
    g = torch.where(torch.isfinite(g), g, torch.zeros_like(g))
    mag = g.abs()
    g = g * torch.where(mag > MAX_MAG, MAX_MAG / (mag + 1e-9), torch.ones_like(mag))
    return g.to(out_dtype)                                       # bk_core.py:210


def sanitise_inputs(he_diag, h0_super, h0_sub):
    """Input clamp + conditional nan_to_num of BKCoreFunction.forward, bk_core.py:256-264."""
    he_diag = torch.clamp(he_diag, -100.0, 100.0)
    h0_super = torch.clamp(h0_super, -10.0, 10.0)
    h0_sub = torch.clamp(h0_sub, -10.0, 10.0)
    if not torch.isfinite(he_diag).all():
        he_diag = torch.nan_to_num(he_diag, nan=0.0, posinf=100.0, neginf=-100.0)
        h0_super = torch.nan_to_num(h0_super, nan=1.0, posinf=10.0, neginf=-10.0)
        h0_sub = torch.nan_to_num(h0_sub, nan=1.0, posinf=10.0, neginf=-10.0)
    return he_diag, h0_super, h0_sub


def forward(he_diag, h0_super, h0_sub, z, phi_k_index: str = "reference"):
    """(features (B,N,2) f32, G (B,N) c64) as BKCoreFunction.forward returns them
    on the sequential path (use_triton=False, use_parallel_scan=False).  bk_core.py:239-300."""
    a, b, c = sanitise_inputs(he_diag, h0_super, h0_sub)
    g = get_diag(a, b, c, z, phi_k_index=phi_k_index)
    feats = torch.stack([g.real, g.imag], dim=-1).to(torch.float32)
    return feats, g


def backward_reference(G, grad_features, grad_G, alpha: float):
    """grad_he_diag of BKCoreFunction.backward, bk_core.py:303-385.

    G: (B,N) c64 saved by forward; grad_features: (B,N,2) or None; grad_G: (B,N) complex or None.
    """
    if grad_features is not None:
        if not torch.isfinite(grad_features).all():
            grad_features = torch.nan_to_num(grad_features)          # bk_core.py:321-325
        gg = torch.complex(grad_features[..., 0], grad_features[..., 1])
    else:
        gg = torch.zeros_like(G)
    if grad_G is not None:
        gg = gg + grad_G
    ggc = gg.conj()
    g2 = G ** 2
    mag = g2.abs()
    den = torch.where(mag < 1e-4, g2 / (mag + 1e-9) * 1e-4, g2)      # bk_core.py:349-356
    theo = -(ggc * g2).real                                          # bk_core.py:359
    h7 = -(ggc / (den + 1e-6)).real                                  # bk_core.py:363
    gv = (1.0 - alpha) * theo + alpha * h7                           # bk_core.py:366-367
    gv = torch.nan_to_num(gv, nan=0.0, posinf=100.0, neginf=-100.0)  # bk_core.py:371-375
    gv = torch.clamp(gv, -100.0, 100.0)                              # bk_core.py:378-380
    return gv.to(torch.float32)


def backward_batched(G, grad_features, alpha: float):
    """grad_he_diag of BatchedAnalyticBKCoreFunction / MemoryOptimizedBKCoreFunction: compute_single_gradient,
    src/models/batched_gradient.py:12-54, on grad_G = complex(grad_features) (:116-119)."""
    gg = torch.complex(grad_features[..., 0], grad_features[..., 1]).to(G.dtype)
    g2 = G ** 2
    mag = g2.abs()
    den = torch.where(mag < 1e-3, g2 / (mag + 1e-9) * 1e-3, g2)      # :33-40
    theo = -(gg * g2).real                                           # :43
    h7 = -(gg / (den + 1e-6)).real                                   # :46
    gv = (1.0 - alpha) * theo + alpha * h7                           # :49
    gv = torch.where(torch.isfinite(gv), gv, torch.zeros_like(gv))   # :52
    return torch.clamp(gv, -1000.0, 1000.0).to(torch.float32)        # :53


def backward_scan_stable(G, grad_output):
    """grad_a of BKScanStableFunction.backward, src/kernels/bk_scan_stable.py:597-598."""
    return (-grad_output.to(G.dtype) * G * G.conj()).real.to(torch.float32)


def analytic_grads(he_diag, h0_super, h0_sub, z, grad_features=None, grad_G=None,
                   phi_k_index: str = "reference"):
    """True gradients (d/da, d/db, d/dc) of L through the regularised, clamped forward,
    by autograd through :func:`get_diag` evaluated in float64/complex128.

    L is defined by its cotangents: dL = sum Re(conj(gbar) dG), gbar = gf[...,0] + i gf[...,1] + grad_G.
    """
    a = he_diag.detach().to(torch.float64).requires_grad_(True)
    b = h0_super.detach().to(torch.float64).requires_grad_(True)
    c = h0_sub.detach().to(torch.float64).requires_grad_(True)
    g = get_diag(a, b, c, z, phi_k_index=phi_k_index, out_dtype=torch.complex128)
    gbar = torch.zeros_like(g)
    if grad_features is not None:
        gbar = gbar + torch.complex(grad_features[..., 0].double(), grad_features[..., 1].double())
    if grad_G is not None:
        gbar = gbar + grad_G.to(torch.complex128)
    loss = (gbar.real * g.real + gbar.imag * g.imag).sum()
    ga, gb, gc = torch.autograd.grad(loss, (a, b, c), allow_unused=True)
    zeros = lambda t, ref: torch.zeros_like(ref) if t is None else t
    return zeros(ga, a), zeros(gb, b), zeros(gc, c)


def dense_resolvent_diag(a, b, c, z):
    """diag((T - zI)^-1) by dense inversion in complex128 (small N only) -- the check the
    reference's own test uses, tests/test_bk_core.py:47-77."""
    B, n = a.shape
    T = torch.zeros(B, n, n, dtype=torch.complex128)
    idx = torch.arange(n)
    T[:, idx, idx] = a.to(torch.complex128) - _as_c128(z, a.device)
    if n > 1:
        T[:, idx[:-1], idx[1:]] = b.to(torch.complex128)
        T[:, idx[1:], idx[:-1]] = c.to(torch.complex128)
    return torch.diagonal(torch.linalg.inv(T), dim1=-2, dim2=-1)
