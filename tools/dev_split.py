"""Development: inspect the per-segment records of the several-workgroups-per-row path."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "project-resnet-bk-an-o-n-language-model-architecture_amd"))
import bkcore
from bkcore import bk_core

def run(B, N, varying, z, conv="reference"):
    g = torch.Generator().manual_seed(17 + B)
    a = -2.0 + 0.5 * torch.randn(B, N, generator=g)
    if varying:
        b = 1.0 + 0.1 * torch.randn(B, N - 1, generator=g); c = 1.0 + 0.1 * torch.randn(B, N - 1, generator=g)
    else:
        b = torch.ones(B, N - 1); c = torch.ones(B, N - 1)
    bk_core._DEBUG["keep_ws"] = True
    bkcore.BKCoreFunction.PHI_K_INDEX = conv
    G = bkcore.vmapped_get_diag(a.cuda(), b.cuda(), c.cuda(), z)
    torch.cuda.synchronize()
    ws = bk_core._DEBUG["last_ws"].cpu().numpy()
    nch = (N + 4095) // 4096
    rf = (4 * B + 255) // 256 * 256
    ls = B * nch * 5 * 256 * 4
    rec = ws[rf + ls:].view(np.float32).reshape(B, -1, 32)
    S = rec.shape[1]
    print(f"B={B} N={N} varying={varying} z={z} conv={conv}: S={S} rowflags={ws[:4*B].view(np.int32)}")
    r = rec[0]
    for s in range(S):
        th_out, ph_out, th0, ph0 = r[s, 0:4], r[s, 5:9], r[s, 10:14], r[s, 15:19]
        cx = lambda v: (v[0] + 1j * v[1]) / (v[2] + 1j * v[3]) if (v[2] != 0 or v[3] != 0) else np.inf
        line = f" seg {s}: bad={r[s,23]:.0f}"
        if s > 0:
            line += f" theta est {cx(th0):.6f} true {cx(r[s-1,0:4]):.6f} rel {abs(cx(th0)-cx(r[s-1,0:4]))/abs(cx(r[s-1,0:4])):.1e}"
        if s < S - 1:
            line += f" | phi est {cx(ph0):.6f} true {cx(r[s+1,5:9]):.6f} rel {abs(cx(ph0)-cx(r[s+1,5:9]))/abs(cx(r[s+1,5:9])):.1e}"
        print(line)

run(2, 65536, False, 1j)
run(2, 65673, True, 0.3 + 0.9j)
run(2, 65673, True, 0.3 + 0.9j, "exact")

# ragged N on the one-workgroup path: rows recomputed?
for N in (1000, 4096 + 7, 32768 + 100):
    g = torch.Generator().manual_seed(N)
    a = (-2.0 + 0.5 * torch.randn(300, N, generator=g)).cuda(); b = torch.ones(300, N - 1).cuda()
    bkcore.vmapped_get_diag(a, b, b, 1j); torch.cuda.synchronize()
    print("one-workgroup path N =", N, "rows recomputed:", bk_core.rows_recomputed())
