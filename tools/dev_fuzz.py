"""Development: random shapes, operand types, conventions and cotangents through the Python mirror against the C oracle
(forward, reference-mode backward) and the float64 adjoint (analytic backward, with and without kept states), on row
subsets.  usage: dev_fuzz.py [cases] [seed]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "project-resnet-bk-an-o-n-language-model-architecture_amd"))
import bkcore
from bkcore import bk_core
from oracle import bk_oracle_c as C
from oracle.bk_analytic_adjoint import adjoint_grads

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = {"fwd": 0.0, "bwd": 0.0, "ana": 0.0, "ana_states": 0.0}
for it in range(cases):
    kind = rng.randint(0, 4)
    if kind == 0:
        B, N = int(rng.randint(1, 40)), int(rng.randint(1, 300))
    elif kind == 1:
        B, N = int(rng.randint(1, 1600)), int(rng.randint(4000, 40000))
    elif kind == 2:
        B, N = int(rng.randint(1, 24)), int(rng.randint(100000, 400000))
    else:
        B, N = int(rng.randint(100, 800)), int(rng.randint(8000, 70000))
    while B * N > (1 << 25):
        B = max(1, B // 2)
    dt = [torch.float32, torch.float32, torch.bfloat16, torch.float16, torch.float64][rng.randint(0, 5)]
    exact = bool(rng.randint(0, 2))
    varying = bool(rng.randint(0, 2))
    z = complex(float(rng.uniform(-0.5, 0.5)), float(rng.uniform(0.7, 1.5)))
    g = torch.Generator().manual_seed(int(rng.randint(0, 1 << 30)))
    a = (-2.0 + 0.5 * torch.randn(B, N, generator=g)).to(dt)
    nb = max(N - 1, 0)
    amp = 0.1 if varying else 0.0
    b = (1.0 + amp * torch.randn(B, nb, generator=g)).to(dt)
    c = (1.0 + amp * torch.randn(B, nb, generator=g)).to(dt)
    ad, bd, cd = a.cuda(), b.cuda(), c.cuda()
    bk_core._DEBUG["keep_ws"] = True
    feats, G = bkcore.bk_forward(ad, bd, cd, z, phi_exact=exact)
    redone = bk_core.rows_recomputed()
    bk_core._DEBUG["keep_ws"] = False
    rows = sorted(set(rng.randint(0, B, size=3).tolist()) | {0, B - 1})
    a64, b64, c64 = a[rows].double().numpy(), b[rows].double().numpy(), c[rows].double().numpy()
    ref = C.forward_f32(a64, b64, c64, z, phi_exact=exact)[1]
    e_f = float(np.max(np.abs(G[rows].cpu().numpy() - ref) / np.abs(ref)))
    gf = torch.randn(B, N, 2, generator=g)
    grad = bkcore.bk_backward_reference(G, gf.cuda(), None, 0.5)
    gref = C.backward_ref(G[rows].cpu().numpy(), gf[rows].numpy(), None, 0.5)
    e_b = float(np.max(np.abs(grad[rows].cpu().numpy() - gref) / (np.abs(gref) + 1e-3 * np.abs(gref).max())))
    e_a = e_s = 0.0
    if dt == torch.float32 and N <= 131072 and N >= 2:
        _, G2, st = bkcore.bk_forward(ad, bd, cd, z, phi_exact=exact, want_states=True)
        ga = bkcore.bk_backward_analytic(ad, bd, cd, z, gf.cuda(), None, phi_exact=exact, G=G2)
        gbar = (gf[rows, :, 0].double() + 1j * gf[rows, :, 1].double()).numpy()
        r3 = adjoint_grads(a64, b64, c64, z, gbar, "exact" if exact else "reference")[:3]
        e_a = max(float(np.abs(x[rows].cpu().numpy() - r).max() / np.abs(r).max()) for x, r in zip(ga, r3))
        if st is not None:
            gs = bkcore.bk_backward_analytic(ad, bd, cd, z, gf.cuda(), None, phi_exact=exact, G=G2, states=st)
            e_s = max(float(np.abs(x[rows].cpu().numpy() - r).max() / np.abs(r).max()) for x, r in zip(gs, r3))
    for k, v in (("fwd", e_f), ("bwd", e_b), ("ana", e_a), ("ana_states", e_s)):
        worst[k] = max(worst[k], v)
    flag = "" if (e_f <= 1e-4 and e_b <= 1e-3 and e_a <= 1e-4 and e_s <= 1e-4 and redone == 0) else "   <-- LOOK"
    print(f"{it:3d} B={B:5d} N={N:7d} {str(dt).split('.')[-1]:8s} exact={int(exact)} varying={int(varying)} z={z:.2f} "
          f"fwd {e_f:.1e} bwd {e_b:.1e} ana {e_a:.1e} ana(states) {e_s:.1e} redone {redone}{flag}", flush=True)
print("worst:", worst, flush=True)
