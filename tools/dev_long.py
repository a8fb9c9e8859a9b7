"""Development: error profile along the row at N = 2^20 against the C oracle.  usage: dev_long.py [--lib path]"""
import argparse, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "project-resnet-bk-an-o-n-language-model-architecture_amd"))
ap = argparse.ArgumentParser(); ap.add_argument("--lib", default=None); ap.add_argument("--B", type=int, default=2)
args = ap.parse_args()
from bkcore import _lib
if args.lib: _lib.LIB_PATH = os.path.abspath(args.lib)
import bkcore
from bkcore import bk_core
from oracle import bk_oracle_c as C
B, N = args.B, 1 << 20
g = torch.Generator().manual_seed(99)
a = -2.0 + 0.5 * torch.randn(B, N, generator=g)
b = torch.ones(B, N - 1)
Gref = C.get_diag(a[:2].numpy(), b[:2].numpy(), b[:2].numpy(), 1j)
bk_core._DEBUG["keep_ws"] = True
G = bkcore.vmapped_get_diag(a.cuda(), b.cuda(), b.cuda(), 1j)[:2].cpu().numpy()
print("rows recomputed", bk_core.rows_recomputed())
rel = (G - Gref) / Gref
for lo in range(0, N, N // 16):
    seg = rel[:, lo:lo + N // 16]
    print(f"[{lo:8d}) max|rel| {np.abs(seg).max():.2e}  mean rel re {seg.real.mean():+.2e} im {seg.imag.mean():+.2e}")
