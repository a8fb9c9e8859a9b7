"""Development A/B: load a given build of the library, check the headline forward against the C oracle on a row
subset, count rows handed to the fallback, time forward and forward+backward.
usage: ab_forward.py [--lib path/to/libbkcore.so] [--tag name]"""
import argparse, os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "project-resnet-bk-an-o-n-language-model-architecture_amd"))
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--tag", default="lib")
ap.add_argument("--iters", type=int, default=100)
ap.add_argument("--dtype", default="f32", choices=["f32", "bf16", "f16"])
ap.add_argument("--rows", type=int, default=1024)
ap.add_argument("--n", type=int, default=32768)
args = ap.parse_args()
from bkcore import _lib
if args.lib:
    _lib.LIB_PATH = os.path.abspath(args.lib)
import bkcore
from bkcore import bk_core
from oracle import bk_oracle_c as C

def timed(fn, iters):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for e0, e1 in ev:
        e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    ts = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
    return ts[len(ts) // 2], ts[0]

out = {"tag": args.tag, "dtype": args.dtype, "rows": args.rows, "n": args.n}
B, N = args.rows, args.n
dt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[args.dtype]
g = torch.Generator(device="cuda").manual_seed(1234)
a = (-2.0 + 0.5 * torch.randn(B, N, device="cuda", generator=g)).to(dt)
b = torch.ones(B, N - 1, device="cuda", dtype=dt)
c = torch.ones(B, N - 1, device="cuda", dtype=dt)     # a buffer of its own: the kernel then reads 12 distinct bytes per element
z = torch.tensor(1j, dtype=torch.complex64, device="cuda")
gf = torch.randn(B, N, 2, device="cuda", generator=g)
bk_core._DEBUG["keep_ws"] = True
G = bkcore.bk_forward(a, b, c, z)[1]
torch.cuda.synchronize()
out["rows_recomputed"] = bk_core.rows_recomputed()
rows = [0, 1, B // 2 - 1, B - 1]
ref = C.forward_f32(a[rows].float().cpu().numpy(), b[rows].float().cpu().numpy(), b[rows].float().cpu().numpy(), 1j)[1]
out["err_headline"] = float(np.max(np.abs(G[rows].cpu().numpy() - ref) / np.abs(ref)))
# position-varying off-diagonals, general z
g2 = torch.Generator().manual_seed(7)
a2 = -2.0 + 0.5 * torch.randn(64, N, generator=g2)
b2 = 1.0 + 0.1 * torch.randn(64, N - 1, generator=g2)
c2 = 1.0 + 0.1 * torch.randn(64, N - 1, generator=g2)
G2 = bkcore.vmapped_get_diag(a2.cuda(), b2.cuda(), c2.cuda(), 0.3 + 0.9j)
torch.cuda.synchronize()
out["rows_recomputed_var"] = bk_core.rows_recomputed()
rows = [0, 31, 63]
ref2 = C.get_diag(a2[rows].numpy(), b2[rows].numpy(), c2[rows].numpy(), 0.3 + 0.9j)
out["err_varying"] = float(np.max(np.abs(G2[rows].cpu().numpy() - ref2) / np.abs(ref2)))
bk_core._DEBUG["keep_ws"] = False
f2_med, f2_min = timed(lambda: bkcore.bk_forward(a, b, c, z), args.iters)
f_med, f_min = timed(lambda: bkcore.bk_forward(a, b, c, z, want_features=False), args.iters)
fb_med, fb_min = timed(lambda: bkcore.bk_backward_reference(bkcore.bk_forward(a, b, c, z, want_features=False)[1], gf, None, 0.5), args.iters)
out.update(fwd2_ms_med=round(f2_med, 4), fwd_ms_med=round(f_med, 4), fwd_ms_min=round(f_min, 4), fwdbwd_ms_med=round(fb_med, 4), fwdbwd_ms_min=round(fb_min, 4),
           fwd_frac=round(20 * B * N / (f_med * 1e-3) / 8e12, 4), fwdbwd_frac=round(40 * B * N / (fb_med * 1e-3) / 8e12, 4))
print(json.dumps(out), flush=True)
