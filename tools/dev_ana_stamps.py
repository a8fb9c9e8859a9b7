"""Development: stage durations of the fused analytic-backward kernel from a BK_SWEEP_STAMPS build."""
import argparse, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "project-resnet-bk-an-o-n-language-model-architecture_amd"))
ap = argparse.ArgumentParser(); ap.add_argument("--lib", required=True)
args = ap.parse_args()
from bkcore import _lib
_lib.LIB_PATH = os.path.abspath(args.lib)
import bkcore
from bkcore import bk_core
B, N, CH = 1024, 32768, 2032
g = torch.Generator(device="cuda").manual_seed(0)
a = -2.0 + 0.5 * torch.randn(B, N, device="cuda", generator=g)
b = 1.0 + 0.1 * torch.randn(B, N - 1, device="cuda", generator=g); c = 1.0 + 0.1 * torch.randn(B, N - 1, device="cuda", generator=g)
gf = torch.randn(B, N, 2, device="cuda", generator=g)
_, G = bkcore.bk_forward(a, b, c, 1j)
bk_core._DEBUG["keep_ws"] = True
for _ in range(3):
    bkcore.bk_backward_analytic(a, b, c, 1j, gf, None, G=G)
torch.cuda.synchronize()
ws = bk_core._DEBUG["last_ana_ws"]
L = _lib.load()
fwd = (int(L.bkcore_workspace_bytes(B, N)) + 255) // 256 * 256
plane = B * N * 8
nch = 1 if N <= 2048 else 1 + (N - 2048 + CH - 1) // CH
st = ws[fwd + plane: fwd + plane + B * nch * 64].view(torch.int64).reshape(B, nch, 8).cpu().numpy().astype(np.float64)
names = ["top", "load+commit+B1", "ratio chains", "B2", "v chains", "B3", "output pass", "stores"]
x = st[:, 1:nch - 1, :]
print("cycles per stage, interior chunks: median / mean")
for k in range(1, 8):
    d = x[:, :, k] - x[:, :, k - 1]
    print(f"   {names[k]:14s} {np.median(d):7.1f} {d.mean():7.1f}")
nx = st[:, 2:nch, 0] - st[:, 1:nch - 1, 7]
tot = st[:, 2:nch, 0] - st[:, 1:nch - 1, 0]
print(f"   -> next top    {np.median(nx):7.1f} {nx.mean():7.1f};  whole chunk {np.median(tot):7.1f} {tot.mean():7.1f}")
rows = st[:, nch - 1, 7] - st[:, 0, 0]
print("row duration us: median %.1f  min %.1f max %.1f" % (np.median(rows) / 100, rows.min() / 100, rows.max() / 100))
