"""Development: many calls of the one-workgroup-per-chunk form at several shapes, with and without another stream
keeping the GPU busy: every call must give the bits of the first, no hand-over may be given up (state word 12), and no
call may take anywhere near the 50 ms a given-up hand-over costs."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "project-resnet-bk-an-o-n-language-model-architecture_amd"))
import bkcore
from bkcore import bk_core

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
side = torch.cuda.Stream()
noise = torch.randn(32 << 20, device="cuda")
z = torch.tensor(1j, dtype=torch.complex64, device="cuda")
for B, N, dt in ((16, 1 << 20, torch.float32), (129, 32768, torch.float32), (600, 32768, torch.float32),
                 (1300, 32768, torch.float32), (700, 131072, torch.bfloat16), (3, 3 * 4096 + 5, torch.float32),
                 (1400, 16384, torch.float16)):
    g = torch.Generator(device="cuda").manual_seed(B + N)
    a = (-2.0 + 0.5 * torch.randn(B, N, device="cuda", generator=g)).to(dt)
    b = torch.ones(B, N - 1, device="cuda", dtype=dt)
    c = torch.ones(B, N - 1, device="cuda", dtype=dt)
    G0 = bkcore.bk_forward(a, b, c, z, want_features=False)[1].clone()
    torch.cuda.synchronize()
    st = bk_core._state_block(a.device, torch.cuda.current_stream().cuda_stream)
    e0 = int(st[9])
    worst, bad, gave_up = 0.0, 0, 0
    t_all = time.perf_counter()
    for i in range(calls):
        if i % 3 == 0:
            with torch.cuda.stream(side):
                noise.mul_(1.0001)
        t0 = time.perf_counter()
        G = bkcore.bk_forward(a, b, c, z, want_features=False)[1]
        if i % 50 == 49:
            torch.cuda.synchronize()
            worst = max(worst, time.perf_counter() - t0)
            bad += 0 if torch.equal(torch.view_as_real(G), torch.view_as_real(G0)) else 1
            gave_up += int(st[12] != 0)
    torch.cuda.synchronize()
    dt_all = time.perf_counter() - t_all
    print(f"B={B} N={N} {str(dt).split('.')[-1]}: {calls} calls in {dt_all:.2f} s, epoch {e0} -> {int(st[9])}, "
          f"checked calls that differ {bad}, gave up {gave_up}, slowest checked call {worst * 1e3:.2f} ms", flush=True)
