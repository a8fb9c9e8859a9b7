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
B, N, CH = 1024, 32768, 2048
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
nch = (N + CH - 1) // CH
st = ws[fwd + plane: fwd + plane + B * 2 * nch * 64].view(torch.int64).reshape(B, 2, nch, 8).cpu().numpy().astype(np.float64)
names = ["top", "commit+B1", "ratio chain", "B2", "v chain", "B3", "output pass", "stores"]
for ph in (0, 1):
    x = st[:, ph, 1:nch - 1, :]
    print("phase", "theta" if ph else "phi", ": ticks (100 MHz) per stage, interior chunks: median / mean")
    for k in range(1, 8):
        d = x[:, :, k] - x[:, :, k - 1]
        print(f"   {names[k]:12s} {np.median(d):7.1f} {d.mean():7.1f}")
    nx = st[:, ph, 2:nch, 0] - st[:, ph, 1:nch - 1, 7]
    tot = st[:, ph, 2:nch, 0] - st[:, ph, 1:nch - 1, 0]
    print(f"   -> next top  {np.median(nx):7.1f} {nx.mean():7.1f};  whole chunk {np.median(tot):7.1f} {tot.mean():7.1f}")
rows = st[:, 1, nch - 1, 7] - st[:, 0, 0, 0]
print("row duration us: median %.1f  min %.1f max %.1f" % (np.median(rows) / 100, rows.min() / 100, rows.max() / 100))
