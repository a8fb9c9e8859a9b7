"""Development: how the workgroups of the headline forward are scheduled (BK_UNIT_STAMPS build): per workgroup start
and end on the 100 MHz wall clock and the CU it ran on.  usage: BKCORE_RELAY_UNITS=S dev_units.py --lib tools/ab/libbkcore_ustamp.so"""
import argparse, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "project-resnet-bk-an-o-n-language-model-architecture_amd"))
ap = argparse.ArgumentParser(); ap.add_argument("--lib", required=True); ap.add_argument("--B", type=int, default=1024)
args = ap.parse_args()
from bkcore import _lib
_lib.LIB_PATH = os.path.abspath(args.lib)
import bkcore
from bkcore import bk_core
B, N = args.B, 32768
S = max(1, int(os.environ.get("BKCORE_RELAY_UNITS", "2")))
g = torch.Generator(device="cuda").manual_seed(1234)
a = -2.0 + 0.5 * torch.randn(B, N, device="cuda", generator=g)
b = torch.ones(B, N - 1, device="cuda")
z = torch.tensor(1j, dtype=torch.complex64, device="cuda")
bk_core._DEBUG["keep_ws"] = True
for _ in range(4):
    bkcore.bk_forward(a, b, b, z, want_features=False)
torch.cuda.synchronize()
ws = bk_core._DEBUG["last_ws"]
U = B * S
off = (4 * B + 255) // 256 * 256 + (8 << 20)
st = ws[off: off + U * 32].view(torch.int64).reshape(U, 4).cpu().numpy()
t0 = st[:, 0].min()
s0 = (st[:, 0] - t0) / 100.0; s1 = (st[:, 1] - t0) / 100.0
hw = st[:, 2]; xcc = (hw >> 32) & 0xF; hwid = hw & 0xFFFFFFFF
cu = (hwid >> 8) & 0xF; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 0x7
cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
print("units", U, "kernel span us", s1.max().round(1), "distinct CUs", len(np.unique(cuid)))
print("start  pct 0/10/25/50/75/90/100:", np.percentile(s0, [0, 10, 25, 50, 75, 90, 100]).round(1))
print("dur    pct 0/10/25/50/75/90/100:", np.percentile(s1 - s0, [0, 10, 25, 50, 75, 90, 100]).round(1))
# occupancy over time: resident workgroups per CU at sample times
ts = np.linspace(0, s1.max(), 21)[:-1]
for tt in ts:
    live = (s0 <= tt) & (s1 > tt)
    cnt = np.bincount(cuid[live].astype(np.int64), minlength=int(cuid.max()) + 1)
    cnt = cnt[np.isin(np.arange(len(cnt)), np.unique(cuid))]
    print("t=%6.1f us  resident %4d   CUs with 0/1/2/3 workgroups: %s" % (tt, live.sum(), np.bincount(cnt, minlength=4)[:5]))
# duration by start order in groups
order = np.argsort(s0)
for k in range(0, U, max(1, U // 8)):
    sel = order[k:k + max(1, U // 8)]
    print("started #%5d..: start %.1f-%.1f us, mean dur %.1f us" % (k, s0[sel].min(), s0[sel].max(), (s1 - s0)[sel].mean()))
