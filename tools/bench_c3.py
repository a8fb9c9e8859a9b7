"""Timing of the long-context configuration (B=16, N=2^20) and a few in-between shapes."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "project-resnet-bk-an-o-n-language-model-architecture_amd"))
if os.environ.get('BKCORE_DEV_LIB'):
    from bkcore import _lib
    _lib.LIB_PATH = os.path.abspath(os.environ['BKCORE_DEV_LIB'])
import bkcore

def t_fwd(B, N, reps=10):
    g = torch.Generator(device="cuda").manual_seed(0)
    a = -2.0 + 0.5 * torch.randn(B, N, device="cuda", generator=g)
    b = torch.ones(B, N - 1, device="cuda")
    for _ in range(3):
        bkcore.BKCoreFunction.apply(a, b, b, 1j)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        bkcore.BKCoreFunction.apply(a, b, b, 1j)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"forward B={B} N={N}: {ms:.3f} ms  {B*N/ms/1e6:.1f} G elems/s... {B*N*28/ms/1e9:.3f} TB/s algorithmic", flush=True)

for B, N in [(16, 1 << 20), (1, 1 << 20), (64, 1 << 18), (128, 1 << 17), (255, 32768), (256, 32768), (1024, 32768)]:
    t_fwd(B, N)
