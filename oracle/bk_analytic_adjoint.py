"""Exact analytic gradient of BK-Core by the adjoint of its ratio recurrences (test infrastructure: the float64
specification of what the HIP analytic backward computes; checked on the CPU against autograd through the oracle).

The forward of the reference (src/models/bk_core.py:96-210), written in ratios of consecutive minors:
    r_0 = a'_0,              r_i = a'_i - k_{i-1}  w(r_{i-1}),   i = 1..N-1      (theta_{i+1}/theta_i)
    s_{N-1} = a'_{N-1},      s_m = a'_m - kphi_m   w(s_{m+1}),   m = N-2..1      (phi_m/phi_{m+1})
    w(r) = exp(g(-ln|r|)) / r,   g(d) = 88 tanh(d/88) - d        (the in-recurrence soft clamp, bk_core.py:60-75,126,172)
    kphi_m = k_{m-1} (reference index, bk_core.py:171) or k_m (exact trailing minors),  k = b c
    G_raw,j = (1/r_j) prod_{m>j} s_m / r_m
    G = clip50(G_raw F(ln|G_raw|)),  F(L) = exp(g(L))                              (bk_core.py:193-208)
With q_j = conj(h_j) G_raw,j (h = cotangent of G_raw), P_m = sum_{j<=m} q_j, rho = dr/r, sigma = ds/s:
    dL = sum_m Re(-P_m rho_m) + sum_{m>=1} Re(P_{m-1} sigma_m)
    rho_i = u_i + kap_i (rho_{i-1} - tau_{i-1} Re rho_{i-1}),  kap_i = k_{i-1} w(r_{i-1}) / r_i,  tau = tanh^2(ln|r|/88)
so the adjoints obey real-linear first-order recurrences
    lam_m = -P_m    + y - tau(r_m) Re y,   y = lam_{m+1} kap_{m+1}          (right -> left)
    mu_m  = P_{m-1} + y - tau(s_m) Re y,   y = mu_{m-1} kap'_{m-1}          (left -> right)
and  grad_a_i = Re[(lam_i / r_i + mu_i / s_i) D_i],  grad_k = -Re[lam_i w(r_{i-1}) / r_i] - Re[mu_m w(s_{m+1}) / s_m].
"""
import numpy as np

K = 88.0


def _g(d):
    return K * np.tanh(d / K) - d


def _w(r):
    """w(r) = exp(g(-ln|r|)) / r"""
    return np.exp(_g(-np.log(np.abs(r)))) / r


def _tau(r):
    return np.tanh(np.log(np.abs(r)) / K) ** 2


def adjoint_grads(a, b, c, z, gbar, phi_k_index="reference"):
    """a (B,N), b,c (B,N-1) float64 (already clamped / sanitised); gbar (B,N) complex128 cotangent of G.
    Returns grad_a, grad_b, grad_c (float64) and G (complex128)."""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64); c = np.asarray(c, np.float64)
    B, N = a.shape
    z = complex(z)
    x = a - z
    mag = np.abs(x)
    f = 1.0 + 1e-3 / (mag + 1e-12)
    ap = x * f
    D = f - 1e-3 * x * x.real / (mag + 1e-12) ** 2 / mag          # d a'/d a for real a
    k = b * c
    off = 1 if phi_k_index == "reference" else 0                  # kphi_m = k[m - off]
    r = np.zeros((B, N), np.complex128)
    s = np.ones((B, N + 1), np.complex128)                        # s[:, 0] unused, s[:, N] = 1 (placeholder)
    r[:, 0] = ap[:, 0]
    for i in range(1, N):
        r[:, i] = ap[:, i] - k[:, i - 1] * _w(r[:, i - 1])
    s[:, N - 1] = ap[:, N - 1]
    for m in range(N - 2, 0, -1):
        s[:, m] = ap[:, m] - k[:, m - off] * _w(s[:, m + 1])
    # G_raw,j = (1/r_j) prod_{m>j} s_m/r_m, in logs
    lr = np.log(r)                                                # complex logs: branch cuts cancel in exp
    ls = np.log(s[:, :N]); ls[:, 0] = 0.0
    tail = np.cumsum((ls - lr)[:, ::-1], axis=1)[:, ::-1]         # sum_{m>=j} (ls_m - lr_m)
    lG = -lr + np.concatenate([tail[:, 1:], np.zeros((B, 1))], axis=1)
    Graw = np.exp(lG)
    L = lG.real
    F = np.exp(_g(L))
    Gc = Graw * F                                                 # before the magnitude clip
    absG = np.abs(Gc)
    clip = absG > 50.0
    G = np.where(clip, Gc * 50.0 / (absG + 1e-9), Gc)
    # cotangent back through the clip, then through F
    gb_ = np.where(clip, (50.0 / absG) * (gbar - Gc * (np.conj(gbar) * Gc).real / absG ** 2), gbar)
    t2 = np.tanh(L / K) ** 2
    h = F * gb_ - t2 * (np.conj(gb_) * Gc).real / np.conj(Graw)
    q = np.conj(h) * Graw
    P = np.cumsum(q, axis=1)
    lam = np.zeros((B, N), np.complex128)
    lam[:, N - 1] = -P[:, N - 1]
    for m in range(N - 2, -1, -1):
        y = lam[:, m + 1] * k[:, m] * _w(r[:, m]) / r[:, m + 1]
        lam[:, m] = -P[:, m] + y - _tau(r[:, m]) * y.real
    mu = np.zeros((B, N), np.complex128)
    if N > 1:
        mu[:, 1] = P[:, 0]
    for m in range(2, N):
        y = mu[:, m - 1] * k[:, m - 1 - off] * _w(s[:, m]) / s[:, m - 1]
        mu[:, m] = P[:, m - 1] + y - _tau(s[:, m]) * y.real
    ga = (lam / r * D).real
    ga[:, 1:] += (mu[:, 1:] / s[:, 1:N] * D[:, 1:]).real
    gk = np.zeros((B, max(N - 1, 0)))
    for i in range(1, N):
        gk[:, i - 1] -= (lam[:, i] * _w(r[:, i - 1]) / r[:, i]).real
    for m in range(1, N - 1):
        gk[:, m - off] -= (mu[:, m] * _w(s[:, m + 1]) / s[:, m]).real
    return ga, c * gk, b * gk, G
