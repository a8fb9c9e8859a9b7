import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "project-resnet-bk-an-o-n-language-model-architecture_amd"))
import bkcore
B, N = 1024, 32768
g = torch.Generator(device="cuda").manual_seed(0)
a = -2.0 + 0.5 * torch.randn(B, N, device="cuda", generator=g)
b = torch.ones(B, N - 1, device="cuda")
gf = torch.randn(B, N, 2, device="cuda", generator=g)
for _ in range(6):
    bkcore.bk_backward_analytic(a, b, b, 1j, gf, None)
torch.cuda.synchronize()
