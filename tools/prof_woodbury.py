"""Development: a few launches of the batched Woodbury correction for rocprofv3 --kernel-trace --stats.
usage: prof_woodbury.py [B N r]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "project-resnet-bk-an-o-n-language-model-architecture_amd"))
import torch
if os.environ.get('BKCORE_DEV_LIB'):
    from bkcore import _lib
    _lib.LIB_PATH = os.path.abspath(os.environ['BKCORE_DEV_LIB'])
import bkcore
B, N, r = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (1024, 32768, 15)
g = torch.Generator(device="cuda").manual_seed(0)
G = 0.4 * torch.complex(torch.randn(B, N, generator=g, device="cuda"), torch.randn(B, N, generator=g, device="cuda"))
amp = 0.5 / N ** 0.25
U, V = amp * torch.randn(N, r, generator=g, device="cuda"), amp * torch.randn(N, r, generator=g, device="cuda")
for _ in range(6):
    bkcore.woodbury_correct(G, U, V)
torch.cuda.synchronize()
