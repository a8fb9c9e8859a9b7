"""Development: the complex-operand kernel (bkcore_get_diag_complex) -- error against the torch oracle and time per
call for a few shapes.  usage: python3 tools/dev_complex.py"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "project-resnet-bk-an-o-n-language-model-architecture_amd"))
sys.path.insert(0, ROOT)
import bkcore
from oracle import bk_oracle_torch as O


def cp(g, shape, m, s, t):
    return torch.complex(m + s * torch.randn(shape, generator=g), t * torch.randn(shape, generator=g))


for B, N, check in ((16, 4096, True), (1024, 4096, False), (64, 32768, False), (1024, 32768, False), (4, 32768, True)):
    g = torch.Generator().manual_seed(N + B)
    a, b, c = cp(g, (B, N), -2.0, 0.5, 0.3), cp(g, (B, N - 1), 1.0, 0.1, 0.2), cp(g, (B, N - 1), 1.0, 0.1, 0.2)
    ad, bd, cd = a.cuda(), b.cuda(), c.cuda()
    G = bkcore.vmapped_get_diag(ad, bd, cd, 0.2 + 0.9j)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        bkcore.vmapped_get_diag(ad, bd, cd, 0.2 + 0.9j)
    e0.record()
    for _ in range(20):
        bkcore.vmapped_get_diag(ad, bd, cd, 0.2 + 0.9j)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    err = ""
    if check:
        rows = slice(0, min(B, 2))
        t0 = time.time()
        ref = O.get_diag(a[rows], b[rows], c[rows], 0.2 + 0.9j).numpy()
        err = f" max rel err {float(np.max(np.abs(G[rows].cpu().numpy() - ref) / np.abs(ref))):.2e} (oracle {time.time() - t0:.1f} s)"
    bytes_ = B * N * 32.0
    print(f"B={B} N={N}: {ms:.4f} ms per call, {bytes_ / ms / 1e6:.0f} GB/s of 32 B/elem{err}", flush=True)
