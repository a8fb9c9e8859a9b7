"""Development: error of the analytic backward against the float64 adjoint as a function of the row length."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "project-resnet-bk-an-o-n-language-model-architecture_amd"))
import bkcore
from oracle.bk_analytic_adjoint import adjoint_grads
for N in [int(x) for x in sys.argv[1:]] or [32768, 131072, 262144, 524288, 1048576]:
    for z in (1j, -0.4 + 0.8j):
        B = 2
        g = torch.Generator().manual_seed(N)
        a = -2.0 + 0.5 * torch.randn(B, N, generator=g)
        b = 1.0 + 0.1 * torch.randn(B, N - 1, generator=g)
        c = 1.0 + 0.1 * torch.randn(B, N - 1, generator=g)
        gf = torch.randn(B, N, 2, generator=g)
        _, G = bkcore.bk_forward(a.cuda(), b.cuda(), c.cuda(), z)
        ga = bkcore.bk_backward_analytic(a.cuda(), b.cuda(), c.cuda(), z, gf.cuda(), None, G=G)
        t0 = time.time()
        gbar = (gf[..., 0].double() + 1j * gf[..., 1].double()).numpy()
        ref = adjoint_grads(a.double().numpy(), b.double().numpy(), c.double().numpy(), z, gbar, "reference")[:3]
        errs = [float(np.abs(x.cpu().numpy() - r).max() / np.abs(r).max()) for x, r in zip(ga, ref)]
        print(f"N={N:8d} z={z!s:12s} analytic err a {errs[0]:.2e} b {errs[1]:.2e} c {errs[2]:.2e}  (oracle {time.time() - t0:.0f} s)", flush=True)
