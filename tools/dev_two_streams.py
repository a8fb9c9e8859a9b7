"""Development: two forwards of the headline shape on two streams at once (each in the two-units-a-row form, whose grid
assumes the GPU to itself): results against the serial ones, wall time, and whether a hand-over was given up."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "project-resnet-bk-an-o-n-language-model-architecture_amd"))
import bkcore
from bkcore import bk_core
B, N = 1024, 32768
g = torch.Generator(device="cuda").manual_seed(0)
xs = []
for i in range(2):
    a = -2.0 + 0.5 * torch.randn(B, N, device="cuda", generator=g)
    xs.append((a, torch.ones(B, N - 1, device="cuda"), torch.ones(B, N - 1, device="cuda")))
z = torch.tensor(1j, dtype=torch.complex64, device="cuda")
ref = [bkcore.bk_forward(*x, z)[1].clone() for x in xs]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for s, x in zip(streams, xs):          # create each stream's state block and workspace
    with torch.cuda.stream(s):
        bkcore.bk_forward(*x, z)
torch.cuda.synchronize()
worst = 0.0
for it in range(200):
    outs = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s, x in zip(streams, xs):
        with torch.cuda.stream(s):
            outs.append(bkcore.bk_forward(*x, z)[1])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    worst = max(worst, dt)
    for o, r in zip(outs, ref):
        if not torch.equal(torch.view_as_real(o), torch.view_as_real(r)):
            err = float((o - r).abs().max() / r.abs().max())
            print(f"iteration {it}: result differs from the serial one by {err:.2e} (took {dt * 1e3:.3f} ms)", flush=True)
print(f"200 concurrent pairs: slowest {worst * 1e3:.3f} ms", flush=True)
