"""Development: the analytic backward started from the forward's boundary states against the one that warms its chains
up again, and against the float64 adjoint specification; timings of both at the headline shape."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "project-resnet-bk-an-o-n-language-model-architecture_amd"))
import bkcore
from bkcore import bk_core
from oracle.bk_analytic_adjoint import adjoint_grads

def rel(x, r):
    return float(np.abs(x - r).max() / np.abs(r).max())

def case(B, N, seed, varying, exact=False, z=1j):
    g = torch.Generator().manual_seed(seed)
    a = -2.0 + 0.5 * torch.randn(B, N, generator=g)
    b = 1.0 + 0.1 * torch.randn(B, N - 1, generator=g) if varying else torch.ones(B, N - 1)
    c = 1.0 + 0.1 * torch.randn(B, N - 1, generator=g) if varying else torch.ones(B, N - 1)
    gf = torch.randn(B, N, 2, generator=g)
    ad, bd, cd, gd = a.cuda(), b.cuda(), c.cuda(), gf.cuda()
    bk_core._DEBUG["keep_ws"] = True
    _, G, st = bkcore.bk_forward(ad, bd, cd, z, raw=True, phi_exact=exact, want_states=True)
    assert st is not None
    g1 = bkcore.bk_backward_analytic(ad, bd, cd, z, gd, None, phi_exact=exact, G=G)
    r1 = bk_core.analytic_rows_recomputed()
    g2 = bkcore.bk_backward_analytic(ad, bd, cd, z, gd, None, phi_exact=exact, G=G, states=st)
    r2 = bk_core.analytic_rows_recomputed()
    bk_core._DEBUG["keep_ws"] = False
    rows = sorted({0, B - 1})
    gbar = (gf[rows, :, 0].double() + 1j * gf[rows, :, 1].double()).numpy()
    ref = adjoint_grads(a[rows].double().numpy(), b[rows].double().numpy(), c[rows].double().numpy(), z, gbar,
                        "exact" if exact else "reference")[:3]
    e1 = max(rel(x[rows].cpu().numpy(), r) for x, r in zip(g1, ref))
    e2 = max(rel(x[rows].cpu().numpy(), r) for x, r in zip(g2, ref))
    d = max(float((x - y).abs().max() / y.abs().max()) for x, y in zip(g2, g1))
    print(f"B={B} N={N} varying={varying} exact={exact}: recompute err {e1:.2e} (rows redone {r1}), states err {e2:.2e} "
          f"(rows redone {r2}), states vs recompute {d:.2e}", flush=True)

import argparse
ap = argparse.ArgumentParser(); ap.add_argument("--time-only", action="store_true"); args = ap.parse_args()
for B, N in () if args.time_only else ((3, 8), (2, 9), (2, 17), (3, 2031), (2, 2032), (2, 2033), (3, 2048), (3, 4095), (2, 4097), (4, 6112), (130, 8192 + 5),
             (5, 40000), (16, 49152 - 7)):
    case(B, N, N + B, True)
if not args.time_only:
    case(3, 5000, 1, False)
    case(3, 5000, 2, True, exact=True, z=0.3 + 0.9j)
    case(2, 32768, 3, True)

def timed(fn, reps=100):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

B, N = 1024, 32768
g = torch.Generator(device="cuda").manual_seed(5)
a = -2.0 + 0.5 * torch.randn(B, N, device="cuda", generator=g)
b = torch.ones(B, N - 1, device="cuda"); c = torch.ones(B, N - 1, device="cuda")
gf = torch.randn(B, N, 2, device="cuda", generator=g)
z = torch.tensor(1j, dtype=torch.complex64, device="cuda")
_, G, st = bkcore.bk_forward(a, b, c, z, want_states=True)
for _ in range(300):                      # bring the device to its working clocks
    bkcore.bk_forward(a, b, c, z)
    bkcore.bk_backward_reference(G, gf, None, 0.5)
t_f = timed(lambda: bkcore.bk_forward(a, b, c, z, share_storage=True))
t_fs = timed(lambda: bkcore.bk_forward(a, b, c, z, share_storage=True, want_states=True))
t_b = timed(lambda: bkcore.bk_backward_analytic(a, b, c, z, gf, None, G=G))
t_bs = timed(lambda: bkcore.bk_backward_analytic(a, b, c, z, gf, None, G=G, states=st))
print(f"headline: forward {t_f:.4f} ms, forward keeping states {t_fs:.4f} ms; analytic backward {t_b:.4f} ms, from states "
      f"{t_bs:.4f} ms; sum {t_f + t_b:.4f} -> {t_fs + t_bs:.4f} ms", flush=True)
