"""Development: phase durations of the ratio-form forward from a BK_RATIO_STAMPS build (s_memtime ticks of wave 0)."""
import argparse, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "project-resnet-bk-an-o-n-language-model-architecture_amd"))
ap = argparse.ArgumentParser(); ap.add_argument("--lib", required=True); ap.add_argument("--chunk", type=int, default=4096)
args = ap.parse_args()
from bkcore import _lib
_lib.LIB_PATH = os.path.abspath(args.lib)
import bkcore
from bkcore import bk_core
B, N = 1024, 32768
g = torch.Generator(device="cuda").manual_seed(1234)
a = -2.0 + 0.5 * torch.randn(B, N, device="cuda", generator=g)
b = torch.ones(B, N - 1, device="cuda")
z = torch.tensor(1j, dtype=torch.complex64, device="cuda")
bk_core._DEBUG["keep_ws"] = True
for _ in range(3):
    bkcore.bk_forward(a, b, b, z)
torch.cuda.synchronize()
ws = bk_core._DEBUG["last_ws"]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    bkcore.bk_forward(a, b, b, z)
e1.record(); torch.cuda.synchronize()
print("forward with stamps: %.4f ms" % (e0.elapsed_time(e1) / 20))
nch = (N + args.chunk - 1) // args.chunk
off = (4 * B + 255) // 256 * 256
full = ws[off: off + B * nch * 16 * 8].view(torch.int64).reshape(B, nch, 16).cpu().numpy()
st = full[:, :, :12]
w0 = full[:, 0, 14].astype(np.float64); w1 = full[:, 0, 15].astype(np.float64)
t00 = w0.min()
w0 = (w0 - t00) / 100.0; w1 = (w1 - t00) / 100.0     # microseconds (100 MHz wall clock)
print("rows: start us percentiles 0/25/50/75/100:", np.percentile(w0, [0, 25, 50, 75, 100]).round(1))
print("rows: duration us percentiles 0/25/50/75/100:", np.percentile(w1 - w0, [0, 25, 50, 75, 100]).round(1), " last end:", w1.max().round(1))
hw = full[:, 0, 13]
xcc = (hw >> 32) & 0xF
hwid = hw & 0xFFFFFFFF
cu = (hwid >> 8) & 0xF; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 0x7
dur = w1 - w0
for x in range(8):
    m = xcc == x
    if m.any():
        print(f"   XCC {x}: {int(m.sum())} rows, duration mean {dur[m].mean():.1f} min {dur[m].min():.1f} max {dur[m].max():.1f}; last end {w1[m].max():.1f}; first-round mean {dur[m & (w0 < 5)].mean():.1f} later {dur[m & (w0 >= 5)].mean():.1f}")
key = xcc * 1000 + se * 100 + sh * 16 + cu
uk = np.unique(key)
percu = np.array([dur[key == k].mean() for k in uk]); cnt = np.array([(key == k).sum() for k in uk])
print(f"   distinct (xcc,se,sh,cu): {len(uk)}; rows per CU min {cnt.min()} max {cnt.max()}; per-CU mean duration min {percu.min():.1f} max {percu.max():.1f}")
for tt in np.linspace(0, w1.max(), 9)[:-1]:
    print(f"   t={tt:7.1f} us: {int(((w0 <= tt) & (w1 > tt)).sum())} rows in flight")
ids = [int(x) for x in os.environ.get("STAMP_IDS", "0,1,2,3,4,5,6,7,8,9,10,11").split(",")]
names = {0: "start", 1: "after B1", 2: "after commit", 3: "after B2", 4: "after warm-up", 5: "after theta", 6: "after phi", 7: "after scan+B3", 8: "after combine", 9: "after epilogue", 10: "after B5", 11: "after stores"}
inter = st[:, 1:nch - 1, :].astype(np.float64)
print("cycles between enabled stamps, interior chunks (median / mean / p90):")
for p_, q_ in zip(ids[:-1], ids[1:]):
    dd = inter[:, :, q_] - inter[:, :, p_]
    print(f"  {names[p_]:16s} -> {names[q_]:16s} {np.median(dd):8.0f} {dd.mean():8.0f} {np.percentile(dd, 90):8.0f}")
nxt = st[:, 0:nch - 2, ids[0]].astype(np.float64) - st[:, 1:nch - 1, ids[-1]].astype(np.float64)
print(f"  {names[ids[-1]]:16s} -> next chunk's {names[ids[0]]:9s} {np.median(nxt):8.0f} {nxt.mean():8.0f}")
per = st[:, 0:nch - 2, ids[0]].astype(np.float64) - st[:, 1:nch - 1, ids[0]].astype(np.float64)
print("chunk period median %.0f cycles" % np.median(per))
row_t = (st[:, 0, 11] - st[:, nch - 1, 0]).astype(np.float64)
print("row total ticks: median %.0f  min %.0f max %.0f;  span of all rows %.0f" % (np.median(row_t), row_t.min(), row_t.max(), st[:, 0, 11].max() - st[:, nch - 1, 0].min()))
start = st[:, nch - 1, 0] - st[:, nch - 1, 0].min()
print("row start ticks percentiles 10/50/90:", np.percentile(start, [10, 50, 90]))
