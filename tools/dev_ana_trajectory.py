"""Development: duration of each of many back-to-back analytic backwards (stream events per launch), to see how the
clock settles; prints the mean of every block of 20 launches.  usage: python3 tools/dev_ana_trajectory.py [launches] [states]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "project-resnet-bk-an-o-n-language-model-architecture_amd"))
import bkcore
B, N = 1024, 32768
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
keep = len(sys.argv) > 2 and sys.argv[2] == "states"
g = torch.Generator(device="cuda").manual_seed(1234)
a = -2.0 + 0.5 * torch.randn(B, N, device="cuda", generator=g)
b = torch.ones(B, N - 1, device="cuda")
c = torch.ones(B, N - 1, device="cuda")
gf = torch.randn(B, N, 2, device="cuda", generator=g)
if keep:
    _, G, st = bkcore.bk_forward(a, b, c, 1j, share_storage=True, want_states=True)
    fn = lambda: bkcore.bk_backward_analytic(a, b, c, 1j, gf, None, G=G, states=st)
else:
    _, G = bkcore.bk_forward(a, b, c, 1j, share_storage=True)
    fn = lambda: bkcore.bk_backward_analytic(a, b, c, 1j, gf, None, G=G)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
torch.cuda.synchronize()
ev[0].record()
for i in range(reps):
    fn()
    ev[i + 1].record()
torch.cuda.synchronize()
d = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(reps)]
print("states" if keep else "plain", "us per launch, blocks of 20:", [round(sum(d[i:i + 20]) / len(d[i:i + 20])) for i in range(0, reps, 20)])
