"""Development A/B: time of the forward that keeps boundary states (and of the plain forward) with a given build.
usage: ab_states_fwd.py [--lib path] [--rows B]"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "project-resnet-bk-an-o-n-language-model-architecture_amd"))
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--rows", type=int, default=1024)
args = ap.parse_args()
from bkcore import _lib
if args.lib:
    _lib.LIB_PATH = os.path.abspath(args.lib)
import bkcore
B, N = args.rows, 32768
g = torch.Generator(device="cuda").manual_seed(1234)
a = -2.0 + 0.5 * torch.randn(B, N, device="cuda", generator=g)
b = torch.ones(B, N - 1, device="cuda")
c = torch.ones(B, N - 1, device="cuda")


def timed(fn, iters=200):
    for _ in range(100):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


t_f = timed(lambda: bkcore.bk_forward(a, b, c, 1j, share_storage=True))
t_s = timed(lambda: bkcore.bk_forward(a, b, c, 1j, share_storage=True, want_states=True))
t_f2 = timed(lambda: bkcore.bk_forward(a, b, c, 1j, share_storage=True))
t_s2 = timed(lambda: bkcore.bk_forward(a, b, c, 1j, share_storage=True, want_states=True))
print(f"{args.lib or 'library'} rows {B}: forward {t_f:.4f} / {t_f2:.4f} ms, keeping states {t_s:.4f} / {t_s2:.4f} ms", flush=True)
