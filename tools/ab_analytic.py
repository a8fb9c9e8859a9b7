"""Development: time and check the analytic backward of one library build.
usage: ab_analytic.py [--lib path/to/libbkcore_x.so]   (checks against oracle/bk_analytic_adjoint.py on small rows)"""
import argparse, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "project-resnet-bk-an-o-n-language-model-architecture_amd"))
ap = argparse.ArgumentParser(); ap.add_argument("--lib"); ap.add_argument("--reps", type=int, default=10)
args = ap.parse_args()
from bkcore import _lib
if args.lib:
    _lib.LIB_PATH = os.path.abspath(args.lib)
import bkcore
from oracle.bk_analytic_adjoint import adjoint_grads
tag = os.path.basename(args.lib) if args.lib else "product"

def rel(x, r):
    return float(np.abs(x.cpu().numpy() - r).max() / max(np.abs(r).max(), 1e-300))
worst = 0.0
for (B, N, z, conv) in [(3, 2049, 1j, "reference"), (2, 6000, 0.3 + 0.5j, "exact"), (2, 20000, 1j, "reference"), (3, 7, 0.5j, "reference")]:
    g = torch.Generator().manual_seed(N)
    a = (-2.0 + 0.5 * torch.randn(B, N, generator=g)).float(); b = (1.0 + 0.1 * torch.randn(B, N - 1, generator=g)).float()
    c = (1.0 + 0.1 * torch.randn(B, N - 1, generator=g)).float(); gf = torch.randn(B, N, 2, generator=g).float()
    ra, rb, rc, _ = adjoint_grads(a.double().numpy(), b.double().numpy(), c.double().numpy(), z, torch.complex(gf[..., 0].double(), gf[..., 1].double()).numpy(), conv)
    _, G = bkcore.bk_forward(a.cuda(), b.cuda(), c.cuda(), z, phi_exact=(conv == "exact"))
    ga, gb, gc = bkcore.bk_backward_analytic(a.cuda(), b.cuda(), c.cuda(), z, gf.cuda(), None, phi_exact=(conv == "exact"), G=G)
    e = max(rel(ga, ra), rel(gb, rb), rel(gc, rc)); worst = max(worst, e)
B, N = 1024, 32768
g = torch.Generator(device="cuda").manual_seed(0)
a = -2.0 + 0.5 * torch.randn(B, N, device="cuda", generator=g)
b = 1.0 + 0.1 * torch.randn(B, N - 1, device="cuda", generator=g); c = 1.0 + 0.1 * torch.randn(B, N - 1, device="cuda", generator=g)
gf = torch.randn(B, N, 2, device="cuda", generator=g)
_, G = bkcore.bk_forward(a, b, c, 1j)
out = []
for want_bc in (True, False):
    for _ in range(3):
        bkcore.bk_backward_analytic(a, b, c, 1j, gf, None, want_bc=want_bc, G=G)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        bkcore.bk_backward_analytic(a, b, c, 1j, gf, None, want_bc=want_bc, G=G)
    e1.record(); torch.cuda.synchronize()
    out.append(e0.elapsed_time(e1) / args.reps)
bkcore.bk_core._DEBUG["keep_ws"] = True
bkcore.bk_backward_analytic(a, b, c, 1j, gf, None, G=G)
nre = bkcore.bk_core.analytic_rows_recomputed()
st = bkcore.bk_core._state_block(a.device, torch.cuda.current_stream().cuda_stream) if hasattr(bkcore.bk_core, "_state_block") else None
print(f"{tag:34s} err {worst:.1e}  bwd {out[0]:.3f} ms  (a only {out[1]:.3f} ms)  rows recomputed {nre}", flush=True)
