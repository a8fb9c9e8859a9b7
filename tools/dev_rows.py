"""Development: forward time as a function of the number of rows (occupancy / tail effects).  usage: dev_rows.py --lib path B1 B2 ..."""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "project-resnet-bk-an-o-n-language-model-architecture_amd"))
ap = argparse.ArgumentParser(); ap.add_argument("--lib", default=None); ap.add_argument("rows", type=int, nargs="+")
args = ap.parse_args()
from bkcore import _lib
if args.lib: _lib.LIB_PATH = os.path.abspath(args.lib)
import bkcore
N = 32768
z = torch.tensor(1j, dtype=torch.complex64, device="cuda")
for B in args.rows:
    g = torch.Generator(device="cuda").manual_seed(1)
    a = -2.0 + 0.5 * torch.randn(B, N, device="cuda", generator=g)
    b = torch.ones(B, N - 1, device="cuda")
    c = torch.ones(B, N - 1, device="cuda")
    for _ in range(5): bkcore.bk_forward(a, b, c, z)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
    for e0, e1 in ev:
        e0.record(); bkcore.bk_forward(a, b, c, z); e1.record()
    torch.cuda.synchronize()
    ts = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
    med = ts[len(ts) // 2]
    print(f"{os.path.basename(args.lib or 'default')}: B={B}: {med:.4f} ms, {med * 1e3 / B:.4f} us/row, {28 * B * N / (med * 1e-3) / 1e12:.2f} TB/s", flush=True)
    del a, b, c
