"""Development: a few analytic backwards at the headline size (for rocprofv3 kernel traces and counters)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "project-resnet-bk-an-o-n-language-model-architecture_amd"))
import bkcore
B, N = 1024, 32768
g = torch.Generator(device="cuda").manual_seed(0)
a = -2.0 + 0.5 * torch.randn(B, N, device="cuda", generator=g)
b = 1.0 + 0.1 * torch.randn(B, N - 1, device="cuda", generator=g)
c = 1.0 + 0.1 * torch.randn(B, N - 1, device="cuda", generator=g)
gf = torch.randn(B, N, 2, device="cuda", generator=g)
_, G = bkcore.bk_forward(a, b, c, 1j, share_storage=True)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    bkcore.bk_backward_analytic(a, b, c, 1j, gf, None, G=G)
torch.cuda.synchronize()
