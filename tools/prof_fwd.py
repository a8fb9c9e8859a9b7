"""Runs the forward kernel a few times on the headline shape (for rocprofv3 counter passes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "project-resnet-bk-an-o-n-language-model-architecture_amd"))
import torch
if os.environ.get('BK_LIB'):
    from bkcore import _lib
    _lib.LIB_PATH = os.path.abspath(os.environ['BK_LIB'])
import bkcore
B, N = int(os.environ.get("BK_B", 1024)), int(os.environ.get("BK_N", 32768))
g = torch.Generator(device="cuda").manual_seed(1234)
a = -2.0 + 0.5 * torch.randn(B, N, generator=g, device="cuda")
b = torch.ones(B, N - 1, device="cuda"); c = torch.ones(B, N - 1, device="cuda")
z = torch.tensor(1j, dtype=torch.complex64, device="cuda")
for _ in range(int(os.environ.get("BK_REPS", 3))):
    bkcore.bk_forward(a, b, c, z, share_storage=os.environ.get('BK_SHARE', '1') == '1')
torch.cuda.synchronize()
