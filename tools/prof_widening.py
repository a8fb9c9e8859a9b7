"""Runs the kernels of the callers around the operator a few times each (for rocprofv3 --kernel-trace --stats):
layer prologue / fused backward / output stage, cumulative sum, Woodbury correction."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "project-resnet-bk-an-o-n-language-model-architecture_amd"))
import torch, bkcore
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
B, N = 1024, 32768
v0 = torch.randn(B, N, generator=g, device=dev)
h0 = torch.full((1, N), -2.0, device=dev)
b = torch.ones(1, N - 1, device=dev)
z = torch.tensor(1j, dtype=torch.complex64, device=dev)
gf = torch.randn(B, N, 2, generator=g, device=dev)
for _ in range(5):
    v = v0.detach().requires_grad_(True)
    f, G = bkcore.FusedBKLayerFunction.apply(v, h0, b, b, z, 5.0, 5.0)
    torch.autograd.backward((f,), (gf,))
n_pos, D = 1 << 20, 256
lin = torch.nn.Linear(2, D).to(dev)
sc = torch.ones(D, device=dev, requires_grad=True)
f0 = torch.randn(n_pos, 2, generator=g, device=dev)
r0 = torch.randn(n_pos, D, generator=g, device=dev)
go = torch.randn(n_pos, D, generator=g, device=dev)
for _ in range(5):
    ff = f0.detach().requires_grad_(True)
    bkcore.FusedSpectralOutputFunction.apply(ff, lin.weight, lin.bias, sc, r0).backward(go)
x = torch.randn(256, 32768, 32, generator=g, device=dev)
xr = torch.randn(4096, 8192, generator=g, device=dev)
for _ in range(5):
    bkcore.fused_associative_scan(x, dim=1)
    bkcore.fused_associative_scan(xr, dim=1)
Gc = 0.4 * torch.complex(torch.randn(B, N, generator=g, device=dev), torch.randn(B, N, generator=g, device=dev))
U = 0.04 * torch.randn(N, 15, generator=g, device=dev)
V = 0.04 * torch.randn(N, 15, generator=g, device=dev)
for _ in range(5):
    bkcore.woodbury_correct(Gc, U, V)
torch.cuda.synchronize()
print("done")
