"""Forward of the long-context configuration (B = 16, N = 2^20) in a loop, for rocprofv3 (kernel trace or counters).
usage: prof_c3.py [--iters K] [--rows B] [--n N]   (BKCORE_DEV_LIB=path selects another build of the library)"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "project-resnet-bk-an-o-n-language-model-architecture_amd"))
if os.environ.get("BKCORE_DEV_LIB"):
    from bkcore import _lib
    _lib.LIB_PATH = os.path.abspath(os.environ["BKCORE_DEV_LIB"])
import bkcore

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--rows", type=int, default=16)
ap.add_argument("--n", type=int, default=1 << 20)
args = ap.parse_args()
B, N = args.rows, args.n
g = torch.Generator(device="cuda").manual_seed(0)
a = -2.0 + 0.5 * torch.randn(B, N, device="cuda", generator=g)
b = torch.ones(B, N - 1, device="cuda")
c = torch.ones(B, N - 1, device="cuda")
z = torch.tensor(1j, dtype=torch.complex64, device="cuda")
for _ in range(5):
    bkcore.BKCoreFunction.apply(a, b, c, z)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(args.iters):
    bkcore.BKCoreFunction.apply(a, b, c, z)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / args.iters
print(f"forward B={B} N={N}: {ms:.4f} ms per call, {B * N * 20 / ms / 1e9:.3f} TB/s at 20 B/elem", flush=True)
