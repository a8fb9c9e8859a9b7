"""Development: forward error of the float32 chains against the C oracle as a function of row length and of how weakly
the ratio maps contract (Im z), to place the length from which rows take the double-precision chains."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "project-resnet-bk-an-o-n-language-model-architecture_amd"))
import bkcore
from bkcore import bk_core
from oracle import bk_oracle_c as C
B = 6
for N in (32768, 49152, 65536, 98304, 131072):
    for z in (1j, 0.8j, 0.7j, -0.4 + 0.8j, 0.45 + 0.75j):
        worst, redone_all = 0.0, 0
        for seed in range(3):
            for varying, exact in ((False, False), (True, False), (True, True)):
                g = torch.Generator().manual_seed(seed * 7 + N)
                a = -2.0 + 0.5 * torch.randn(B, N, generator=g)
                amp = 0.1 if varying else 0.0
                b = 1.0 + amp * torch.randn(B, N - 1, generator=g)
                c = 1.0 + amp * torch.randn(B, N - 1, generator=g)
                bk_core._DEBUG["keep_ws"] = True
                _, G = bkcore.bk_forward(a.cuda(), b.cuda(), c.cuda(), z, phi_exact=exact)
                redone_all += bk_core.rows_recomputed()
                bk_core._DEBUG["keep_ws"] = False
                ref = C.forward_f32(a.numpy(), b.numpy(), c.numpy(), z, phi_exact=exact)[1]
                worst = max(worst, float(np.max(np.abs(G.cpu().numpy() - ref) / np.abs(ref))))
        print(f"N={N:7d} z={z!s:14s} worst err {worst:.2e}  rows redone {redone_all} of {9 * B}", flush=True)
