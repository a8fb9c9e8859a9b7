"""Development: a few analytic backwards at the headline size (for rocprofv3 kernel traces and counters), on the very
inputs bench.py times (he_diag = -2 + 0.5 randn with seed 1234, h0_super = h0_sub = 1, z = 1j), so that a trace of this
script and bench.py's roofline_analytic speak of the same launches.

    python3 tools/prof_ana.py [launches=5] [plain|states|varied]

plain: the backward recomputes the ratio chains' warm-ups.  states: it starts from the boundary states the forward kept
(the default of analytic mode).  varied: plain, with off-diagonals 1 + 0.1 randn (more clipped elements, slower)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "project-resnet-bk-an-o-n-language-model-architecture_amd"))
import bkcore
B, N = 1024, 32768
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
kind = sys.argv[2] if len(sys.argv) > 2 else "plain"
assert kind in ("plain", "states", "varied")
g = torch.Generator(device="cuda").manual_seed(1234)
a = -2.0 + 0.5 * torch.randn(B, N, device="cuda", generator=g)
b = torch.ones(B, N - 1, device="cuda")
c = torch.ones(B, N - 1, device="cuda")
gf = torch.randn(B, N, 2, device="cuda", generator=g)
if kind == "varied":
    b = 1.0 + 0.1 * torch.randn(B, N - 1, device="cuda", generator=g)
    c = 1.0 + 0.1 * torch.randn(B, N - 1, device="cuda", generator=g)
if kind == "states":
    _, G, st = bkcore.bk_forward(a, b, c, 1j, share_storage=True, want_states=True)
    assert st is not None
    for _ in range(reps):
        bkcore.bk_backward_analytic(a, b, c, 1j, gf, None, G=G, states=st)
else:
    _, G = bkcore.bk_forward(a, b, c, 1j, share_storage=True)
    for _ in range(reps):
        bkcore.bk_backward_analytic(a, b, c, 1j, gf, None, G=G)
torch.cuda.synchronize()
