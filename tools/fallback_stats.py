import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "project-resnet-bk-an-o-n-language-model-architecture_amd"))
import torch, bkcore
from bkcore import bk_core
bk_core._DEBUG["keep_ws"] = True
def run(name, a, b, c, z):
    bkcore.bk_forward(a, b, c, z)
    torch.cuda.synchronize()
    print(f"{name}: rows recomputed {bk_core.rows_recomputed()} / {a.shape[0]}")
g = torch.Generator(device="cuda").manual_seed(1)
B, N = 256, 32768
run("laplacian-like z=1j", -2 + 0.5 * torch.randn(B, N, generator=g, device="cuda"), torch.ones(B, N - 1, device="cuda"), torch.ones(B, N - 1, device="cuda"), 1j)
run("laplacian-like varying bc", -2 + 0.5 * torch.randn(B, N, generator=g, device="cuda"), 1 + 0.1 * torch.randn(B, N - 1, generator=g, device="cuda"), 1 + 0.1 * torch.randn(B, N - 1, generator=g, device="cuda"), 1j)
run("diag-only model-like", 1 + 5 * torch.tanh(torch.randn(B, N, generator=g, device="cuda") / 5), torch.zeros(B, N - 1, device="cuda"), torch.zeros(B, N - 1, device="cuda"), 1.3133j)
run("bench randn z=0.1+0.1j", torch.randn(B, N, generator=g, device="cuda"), torch.randn(B, N - 1, generator=g, device="cuda"), torch.randn(B, N - 1, generator=g, device="cuda"), 0.1 + 0.1j)
run("laplacian z=0.1j (weakly damped)", -2 + 0.5 * torch.randn(B, N, generator=g, device="cuda"), torch.ones(B, N - 1, device="cuda"), torch.ones(B, N - 1, device="cuda"), 0.1j)
run("a=0 b=c=1 z=0.05j (in-band)", torch.zeros(B, N, device="cuda"), torch.ones(B, N - 1, device="cuda"), torch.ones(B, N - 1, device="cuda"), 0.05j)
