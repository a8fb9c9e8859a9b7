"""NumPy model of the GPU forward algorithm (development aid and CPU-checkable specification).

Test infrastructure only.  It executes, in float32/complex64 arithmetic with integer power-of-two
exponents, the same steps the HIP forward kernel performs, so the algorithm itself (two-pass
fixed point for the in-recurrence soft clamp, scaled 2x2 transfer-matrix scan, epilogue) can be
checked against the oracle on the CPU:

  per direction (theta: left->right, phi: right->left), per chunk of T lanes x E elements
    pass 1: lane aggregates of M_p = [[alpha_p, -kappa_p],[1,0]]  -> exclusive scan over lanes
            -> re-run from each lane's incoming state, giving d_p = ln|prev2| - ln|prev|
            -> w_p = exp(88 tanh(d_p/88) - d_p)    (the reference's "rubber wall", bk_core.py:60-75,127-136,
                                                     expressed as a correction to the exact exp(d_p))
    pass 2: the same scan with kappa'_p = kappa_p * w_p  -> incoming state per element (mantissa, exponent)
  epilogue: G_raw = theta_i phi_i / theta_N ; L = ln|G_raw| ; G = G_raw * exp(88 tanh(L/88) - L)
            (bk_core.py:190-201), non-finite -> 0, |G| > 50 -> 50 G/(|G|+1e-9) (bk_core.py:204-208)
"""
import numpy as np

F = np.float32
C64 = np.complex64


def _regularise_f32(x):
    """a' = x + 1e-3 x/(|x|+1e-12)  (bk_core.py:78-94) in float32."""
    mag = np.abs(x).astype(F)
    scale = (F(1.0) + F(1e-3) / (mag + F(1e-12))).astype(F)
    out = (x * scale).astype(C64)
    return np.where(mag < 1e-12, x + F(1e-3), out).astype(C64)


def _norm_exp(maxabs):
    """exponent e such that maxabs * 2^-e in [0.5, 1); 0 for maxabs == 0."""
    _, e = np.frexp(maxabs)
    return np.where(maxabs > 0, e, 0).astype(np.int32)


def _maxabs(*zs):
    m = None
    for z in zs:
        v = np.maximum(np.abs(z.real), np.abs(z.imag))
        m = v if m is None else np.maximum(m, v)
    return m


def _ld(z, e):
    return (np.ldexp(z.real, e) + 1j * np.ldexp(z.imag, e)).astype(C64)


class Mat:
    """2x2 complex64 mantissa matrices with a shared int32 exponent, batched over leading dims."""

    def __init__(self, m00, m01, m10, m11, e):
        self.m = [m00.astype(C64), m01.astype(C64), m10.astype(C64), m11.astype(C64)]
        self.e = e.astype(np.int32)

    @staticmethod
    def identity(shape):
        o = np.ones(shape, C64)
        z = np.zeros(shape, C64)
        return Mat(o, z, z, o.copy(), np.zeros(shape, np.int32))

    def normalise(self):
        de = _norm_exp(_maxabs(*self.m))
        self.m = [_ld(x, -de) for x in self.m]
        self.e = self.e + de
        return self

    def index(self, idx):
        return Mat(*(x[idx] for x in self.m), self.e[idx])


def matmul(A, B):
    """A @ B (A applied after B), float32 arithmetic, renormalised."""
    a, b = A.m, B.m
    r = Mat((a[0] * b[0] + a[1] * b[2]), (a[0] * b[1] + a[1] * b[3]),
            (a[2] * b[0] + a[3] * b[2]), (a[2] * b[1] + a[3] * b[3]), A.e + B.e)
    return r.normalise()


def _lane_aggregate(alpha, kappa, valid):
    """alpha,kappa,valid: [..., E] in processing order.  Returns Mat = M_{E-1}...M_0 (identity for invalid)."""
    shp = alpha.shape[:-1]
    A = Mat.identity(shp)
    E = alpha.shape[-1]
    for e in range(E):
        al, ka, va = alpha[..., e], kappa[..., e], valid[..., e]
        n00 = (al * A.m[0] - ka * A.m[2]).astype(C64)
        n01 = (al * A.m[1] - ka * A.m[3]).astype(C64)
        A = Mat(np.where(va, n00, A.m[0]), np.where(va, n01, A.m[1]),
                np.where(va, A.m[0], A.m[2]), np.where(va, A.m[1], A.m[3]), A.e)
        if e % 4 == 3 or e == E - 1:
            A.normalise()
    return A


def _exclusive_scan_lanes(A):
    """Kogge-Stone inclusive scan over the lane axis (-1 of the leading dims), then shift to exclusive."""
    T = A.e.shape[-1]
    inc = A
    s = 1
    while s < T:
        # combine lane t with lane t-s: new = inc[t] @ inc[t-s]
        left = Mat(*(np.concatenate([np.zeros_like(x[..., :s]), x[..., :-s]], -1) for x in inc.m),
                   np.concatenate([np.zeros_like(inc.e[..., :s]), inc.e[..., :-s]], -1))
        comb = matmul(inc, left)
        take = (np.arange(T) >= s)
        inc = Mat(*(np.where(take, c, x) for c, x in zip(comb.m, inc.m)), np.where(take, comb.e, inc.e))
        s *= 2
    ident = Mat.identity(A.e.shape[:-1] + (1,))
    exc = Mat(*(np.concatenate([i, x[..., :-1]], -1) for i, x in zip(ident.m, inc.m)),
              np.concatenate([ident.e, inc.e[..., :-1]], -1))
    total = inc.index((Ellipsis, -1))
    return exc, total


def _matvec(P, x0, x1, ex):
    y0 = (P.m[0] * x0 + P.m[1] * x1).astype(C64)
    y1 = (P.m[2] * x0 + P.m[3] * x1).astype(C64)
    e = P.e + ex
    de = _norm_exp(_maxabs(y0, y1))
    return _ld(y0, -de), _ld(y1, -de), e + de


def _wall_correction(d):
    """exp(88 tanh(d/88) - d) in float32; series for |d| <= 8, direct otherwise."""
    d = d.astype(F)
    u = (d / F(88.0)).astype(F)
    s = (u * u).astype(F)
    g_small = (-F(88.0) * u * s * (F(1 / 3) - s * (F(2 / 15) - s * F(17 / 315)))).astype(F)
    with np.errstate(over="ignore", invalid="ignore"):
        g_big = (F(88.0) * np.tanh(u) - d).astype(F)
    g = np.where(np.abs(d) <= 8.0, g_small, g_big)
    return np.exp(g).astype(F)


def _rerun(alpha, kappa, valid, x0, x1, ex, want_wall):
    """Serial re-run over the lane's E elements from incoming state (x0,x1)*2^ex.
    Returns per-element incoming first component (mantissa, exponent) and, if want_wall, w per element."""
    E = alpha.shape[-1]
    out_m = np.zeros(alpha.shape, C64)
    out_e = np.zeros(alpha.shape, np.int32)
    w = np.ones(alpha.shape, F)
    for e in range(E):
        out_m[..., e] = x0
        out_e[..., e] = ex
        if want_wall:
            n0 = (x0.real * x0.real + x0.imag * x0.imag).astype(F)
            n1 = (x1.real * x1.real + x1.imag * x1.imag).astype(F)
            with np.errstate(divide="ignore", invalid="ignore"):
                d = (F(0.5) * (np.log(n1) - np.log(n0))).astype(F)
            ok = (n1 > 0) & (n0 > 0) & np.isfinite(d)
            w[..., e] = np.where(ok, _wall_correction(np.where(ok, d, 0)), F(1.0))
        al, ka, va = alpha[..., e], kappa[..., e], valid[..., e]
        y0 = (al * x0 - ka * x1).astype(C64)
        x1 = np.where(va, x0, x1)
        x0 = np.where(va, y0, x0)
        de = _norm_exp(_maxabs(x0, x1))
        x0, x1, ex = _ld(x0, -de), _ld(x1, -de), ex + de
    return out_m, out_e, w, (x0, x1, ex)


def directional_scan(alpha, kappa, T=64, E=8):
    """alpha (c64), kappa (f32): [B, N] in processing order.  Returns per-position incoming first component
    (mantissa [B,N] c64, exponent [B,N] i32) of the pass-2 (wall-corrected) recurrence and the final state."""
    B, N = alpha.shape
    Cn = T * E
    nch = (N + Cn - 1) // Cn
    pad = nch * Cn - N
    al = np.concatenate([alpha, np.ones((B, pad), C64)], 1).reshape(B, nch, T, E)
    ka = np.concatenate([kappa, np.zeros((B, pad), F)], 1).reshape(B, nch, T, E)
    va = (np.arange(nch * Cn) < N).reshape(1, nch, T, E) & np.ones((B, 1, 1, 1), bool)
    om = np.zeros((B, nch, T, E), C64)
    oe = np.zeros((B, nch, T, E), np.int32)
    # running chunk carries for pass 1 (linear) and pass 2 (wall-corrected)
    c1 = (np.ones(B, C64), np.zeros(B, C64), np.zeros(B, np.int32))
    c2 = (np.ones(B, C64), np.zeros(B, C64), np.zeros(B, np.int32))
    for ch in range(nch):
        a_, k_, v_ = al[:, ch], ka[:, ch], va[:, ch]
        # pass 1
        A = _lane_aggregate(a_, k_, v_)
        exc, tot = _exclusive_scan_lanes(A)
        x0, x1, ex = _matvec(exc, c1[0][:, None], c1[1][:, None], c1[2][:, None])
        _, _, w, _ = _rerun(a_, k_, v_, x0, x1, ex, True)
        c1 = _matvec(tot, *c1)
        # pass 2
        k2 = (k_ * w).astype(F)
        A = _lane_aggregate(a_, k2, v_)
        exc, tot = _exclusive_scan_lanes(A)
        x0, x1, ex = _matvec(exc, c2[0][:, None], c2[1][:, None], c2[2][:, None])
        m_, e_, _, _ = _rerun(a_, k2, v_, x0, x1, ex, False)
        om[:, ch], oe[:, ch] = m_, e_
        c2 = _matvec(tot, *c2)
    return om.reshape(B, -1)[:, :N], oe.reshape(B, -1)[:, :N], c2


def forward_model(a, b, c, z, phi_exact=False, T=64, E=8, wall=True):
    """float32 model of the GPU forward: returns G complex64 [B,N] (no input clamps)."""
    a = np.asarray(a, F)
    B, N = a.shape
    if N == 0:
        return np.zeros((B, 0), C64)
    z = C64(z)
    x = (a - z).astype(C64)
    ap = _regularise_f32(x)
    k = (np.asarray(b, F) * np.asarray(c, F)).astype(F)                    # k_i, i = 0..N-2
    kprev = np.concatenate([np.zeros((B, 1), F), k], 1)                    # k_{i-1} per element
    knext = np.concatenate([k, np.zeros((B, 1), F)], 1)                    # k_i per element (exact phi)
    th_m, th_e, det = directional_scan(ap, kprev, T, E)
    kphi = (knext if phi_exact else kprev)[:, ::-1]
    ph_m, ph_e, _ = directional_scan(ap[:, ::-1].copy(), kphi.copy(), T, E)
    ph_m, ph_e = ph_m[:, ::-1], ph_e[:, ::-1]
    # epilogue
    det_m, det_e = det[0], det[2]
    with np.errstate(all="ignore"):
        inv = (C64(1.0) / det_m).astype(C64)
        P = (th_m * ph_m).astype(C64) * inv[:, None]
        P = P.astype(C64)
        Eexp = th_e + ph_e - det_e[:, None]
        n = (P.real * P.real + P.imag * P.imag).astype(F)
        L = (F(0.5 * np.log(2.0)) * (np.log2(n).astype(F) + F(2.0) * Eexp.astype(F))).astype(F)
        fast = np.abs(L) < 8.0
        corr = _wall_correction(np.where(fast, L, 0))
        Gf = _ld((P * corr).astype(C64), np.where(fast, Eexp, 0))
        mag_slow = np.exp(F(88.0) * np.tanh(L / F(88.0))).astype(F)
        Gs = (P / np.sqrt(n).astype(F) * mag_slow).astype(C64)
        G = np.where(fast, Gf, Gs)
        G = np.where(np.isfinite(G.real) & np.isfinite(G.imag), G, 0).astype(C64)
        mag = np.abs(G).astype(F)
        G = np.where(mag > 50.0, G * (F(50.0) / (mag + F(1e-9))), G).astype(C64)
    return G
