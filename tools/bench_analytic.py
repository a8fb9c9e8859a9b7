"""Timing of the analytic backward at the headline size, and its error on the golden fixtures."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "project-resnet-bk-an-o-n-language-model-architecture_amd"))
import bkcore

d = np.load(os.path.join(ROOT, "tests/golden/bkcore_golden.npz"))
for n in [str(s) for s in d["__names__"]]:
    if f"{n}/grad_true_a" not in d.files:
        continue
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).float().cuda()
    a = dev(np.clip(d[f"{n}/a"].astype(np.float64), -100, 100)); b = dev(np.clip(d[f"{n}/b"].astype(np.float64), -10, 10))
    c = dev(np.clip(d[f"{n}/c"].astype(np.float64), -10, 10))
    ga, gb, gc = bkcore.bk_backward_analytic(a, b, c, complex(*d[f"{n}/z"]), dev(d[f"{n}/gfeat"]), None)
    rel = lambda x, r: 0.0 if r.size == 0 else float(np.abs(x.cpu().numpy() - r).max() / max(np.abs(r).max(), 1e-300))
    print(f"{n:22s} ga {rel(ga, d[f'{n}/grad_true_a']):.1e} gb {rel(gb, d[f'{n}/grad_true_b']):.1e} gc {rel(gc, d[f'{n}/grad_true_c']):.1e}", flush=True)

B, N = 1024, 32768
g = torch.Generator(device="cuda").manual_seed(0)
a = -2.0 + 0.5 * torch.randn(B, N, device="cuda", generator=g)
b = torch.ones(B, N - 1, device="cuda"); c = torch.ones(B, N - 1, device="cuda")
gf = torch.randn(B, N, 2, device="cuda", generator=g)
_, G = bkcore.bk_forward(a, b, c, 1j)
for want_bc, Gs in ((True, G), (True, None), (False, G)):
    for _ in range(3):
        bkcore.bk_backward_analytic(a, b, c, 1j, gf, None, want_bc=want_bc, G=Gs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        bkcore.bk_backward_analytic(a, b, c, 1j, gf, None, want_bc=want_bc, G=Gs)
    e1.record(); torch.cuda.synchronize()
    print(f"analytic backward B={B} N={N} want_bc={want_bc} G={'saved' if Gs is not None else 'recomputed'}: "
          f"{e0.elapsed_time(e1)/10:.3f} ms", flush=True)
