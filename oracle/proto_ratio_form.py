"""Float32 numerical model of a single-sweep ratio formulation of the forward (development aid, test infrastructure):
r_j = theta_{j+1}/theta_j and s_m = phi_m/phi_{m+1} by their own first-order recurrences (bk_core.py:120-184 in ratio
form), G_raw,j = (1/r_j) prod_{m>j} s_m/r_m with the products carried as complex64 mantissa x 2^e.
Prints the error against the C oracle."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import bk_oracle_c as C

f32 = np.float32
K = f32(88.0)


def g32(d):
    d = d.astype(f32)
    return (K * np.tanh(d / K, dtype=f32) - d).astype(f32) if False else (np.float64(88.0) * np.tanh(d.astype(np.float64) / 88.0) - d).astype(f32)


def model(a, b, c, z, exact=False, block=16):
    B, N = a.shape
    x = (a.astype(f32) - np.complex64(z)).astype(np.complex64)
    mag = np.abs(x).astype(f32)
    al = (x * (f32(1) + f32(1e-3) / mag)).astype(np.complex64)
    k = (b.astype(f32) * c.astype(f32)).astype(f32)
    r = np.zeros((B, N), np.complex64)
    s = np.ones((B, N + 1), np.complex64)
    r[:, 0] = al[:, 0]
    for i in range(1, N):
        rp = r[:, i - 1]
        lnr = (f32(0.5) * np.log((rp.real ** 2 + rp.imag ** 2).astype(f32))).astype(f32)
        w = np.exp(g32(-lnr)).astype(f32)
        r[:, i] = (al[:, i] - (k[:, i - 1] * w) / rp).astype(np.complex64)
    s[:, N - 1] = al[:, N - 1]
    off = 0 if exact else 1
    for m in range(N - 2, 0, -1):
        sp = s[:, m + 1]
        lns = (f32(0.5) * np.log((sp.real ** 2 + sp.imag ** 2).astype(f32))).astype(f32)
        w = np.exp(g32(-lns)).astype(f32)
        s[:, m] = (al[:, m] - (k[:, m - off] * w) / sp).astype(np.complex64)
    q = (s[:, :N] / r).astype(np.complex64)
    q[:, 0] = 1.0
    # suffix products P_j = prod_{m>j} q_m as (mantissa, exponent), blocked like a scan: products inside blocks of
    # `block`, then a sequential pass over the block totals
    nb = (N + block - 1) // block
    pad = nb * block - N
    qp = np.concatenate([q, np.ones((B, pad), np.complex64)], 1).reshape(B, nb, block)
    inner = np.ones((B, nb, block), np.complex64)        # inner[.., e] = prod_{e' > e in block} q
    run = np.ones((B, nb), np.complex64)
    for e in range(block - 1, -1, -1):
        inner[:, :, e] = run
        run = (run * qp[:, :, e]).astype(np.complex64)
    tot_m = np.ones((B, nb), np.complex64); tot_e = np.zeros((B, nb), np.int64)
    cm = np.ones(B, np.complex64); ce = np.zeros(B, np.int64)
    for blk in range(nb - 1, -1, -1):
        tot_m[:, blk] = cm; tot_e[:, blk] = ce
        cm = (cm * run[:, blk]).astype(np.complex64)
        mx = np.maximum(np.abs(cm.real), np.abs(cm.imag))
        _, ex = np.frexp(mx)
        cm = (cm * np.ldexp(f32(1), -ex).astype(f32)).astype(np.complex64)
        ce = ce + ex
    Pm = (inner * tot_m[:, :, None]).astype(np.complex64).reshape(B, -1)[:, :N]
    Pe = np.repeat(tot_e, block, axis=1)[:, :N]
    Graw = (Pm / r).astype(np.complex64).astype(np.complex128) * np.exp2(Pe.astype(np.float64))
    L = np.log(np.abs(Graw))
    G = Graw * np.exp(88.0 * np.tanh(L / 88.0) - L)
    return G


if __name__ == "__main__":
    for N in (1024, 32768, 1 << 18):
        rng = np.random.default_rng(N)
        B = 4
        a = (-2.0 + 0.5 * rng.standard_normal((B, N))).astype(np.float32)
        b = (1.0 + 0.1 * rng.standard_normal((B, N - 1))).astype(np.float32)
        c = (1.0 + 0.1 * rng.standard_normal((B, N - 1))).astype(np.float32)
        z = 0.3 + 0.9j
        G = model(a, b, c, z)
        ref = C.get_diag(a, b, c, z, want128=True)
        ref = ref[1] if isinstance(ref, tuple) else ref
        err = np.abs(G - ref) / np.abs(ref)
        print(f"N={N}: max rel err {err.max():.2e}  median {np.median(err):.2e}", flush=True)


def debug():
    N = 1 << 18
    rng = np.random.default_rng(N)
    B = 4
    a = (-2.0 + 0.5 * rng.standard_normal((B, N))).astype(np.float32)
    b = (1.0 + 0.1 * rng.standard_normal((B, N - 1))).astype(np.float32)
    c = (1.0 + 0.1 * rng.standard_normal((B, N - 1))).astype(np.float32)
    z = 0.3 + 0.9j
    G = model(a, b, c, z)
    ref = C.get_diag(a, b, c, z, want128=True)
    ref = ref[1] if isinstance(ref, tuple) else ref
    err = np.abs(G - ref) / np.abs(ref)
    bad = np.argwhere(err > 1e-3)
    print("bad count", len(bad), bad[:10])
    for (r0, j) in bad[:3]:
        print(r0, j, G[r0, j], ref[r0, j], "neighbours err", err[r0, max(0, j - 2):j + 3])
    print("max err excluding bad", err[err <= 1e-3].max())
