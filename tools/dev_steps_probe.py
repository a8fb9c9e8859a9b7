"""Development: per-step stream-event times of the benchmark's autograd step right after a process has started
(kernel-only warm-up, five steps, then the timed ones): how long the host path takes to reach steady state."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "project-resnet-bk-an-o-n-language-model-architecture_amd"))
import bkcore
from bkcore import BKCoreFunction
B, N = 1024, 32768
g = torch.Generator(device="cuda").manual_seed(1234)
a = (-2.0 + 0.5 * torch.randn(B, N, device="cuda", generator=g)).requires_grad_(True)
b = torch.ones(B, N - 1, device="cuda"); c = torch.ones(B, N - 1, device="cuda")
z = torch.tensor(1j, dtype=torch.complex64, device="cuda")
gfeat = torch.randn(B, N, 2, device="cuda", generator=g)
def step():
    a.grad = None
    feats, G = BKCoreFunction.apply(a, b, c, z)
    torch.autograd.backward((feats,), (gfeat,))
ad = a.detach()
with torch.no_grad():
    for _ in range(200):
        Gw = bkcore.bk_forward(ad, b, c, z, share_storage=True)[1]
        bkcore.bk_backward_reference(Gw, gfeat, None, 0.5)
torch.cuda.synchronize()
for _ in range(5): step()
torch.cuda.synchronize()
for trial in range(4):
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(20):
        step(); ev[i + 1].record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("host total/20 = %.4f ms; per-step event ms:" % (dt / 20 * 1e3), [round(ev[i].elapsed_time(ev[i + 1]), 3) for i in range(20)])
