"""Host time per forward+backward call of the operator (tiny input, so the GPU is never the bottleneck)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "project-resnet-bk-an-o-n-language-model-architecture_amd"))
import torch, bkcore
from bkcore import BKCoreFunction
a = torch.randn(2, 64, device="cuda").requires_grad_(True)
b = torch.ones(2, 63, device="cuda")
z = torch.tensor(1j, dtype=torch.complex64, device="cuda")
gf = torch.randn(2, 64, 2, device="cuda")
def step():
    a.grad = None
    f, G = BKCoreFunction.apply(a, b, b, z)
    torch.autograd.backward((f,), (gf,))
for _ in range(50): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000): step()
torch.cuda.synchronize()
print(f"fwd+bwd through autograd: {(time.perf_counter()-t0)/2000*1e6:.1f} us per step")
ad = a.detach()
with torch.no_grad():
    for _ in range(50): bkcore.bk_forward(ad, b, b, z)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000): bkcore.bk_forward(ad, b, b, z)
    torch.cuda.synchronize()
    print(f"bk_forward alone: {(time.perf_counter()-t0)/2000*1e6:.1f} us per call")
