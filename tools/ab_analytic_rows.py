"""Development: time of the analytic backward against the row count (event-timed).
usage: ab_analytic_rows.py N rows...   (BKCORE_SWEEP_ROWS_MAX / BKCORE_SWEEP_MIN_CHUNKS: development overrides of the
segment plan, read by the library)"""
import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "project-resnet-bk-an-o-n-language-model-architecture_amd"))
import bkcore
N = int(sys.argv[1])
for B in map(int, sys.argv[2:]):
    g = torch.Generator(device="cuda").manual_seed(0)
    a = -2.0 + 0.5 * torch.randn(B, N, device="cuda", generator=g)
    b = 1.0 + 0.1 * torch.randn(B, N - 1, device="cuda", generator=g)
    c = 1.0 + 0.1 * torch.randn(B, N - 1, device="cuda", generator=g)
    gf = torch.randn(B, N, 2, device="cuda", generator=g)
    _, G = bkcore.bk_forward(a, b, c, 1j, share_storage=True)
    for _ in range(5):
        out = bkcore.bk_backward_analytic(a, b, c, 1j, gf, None, G=G)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        out = bkcore.bk_backward_analytic(a, b, c, 1j, gf, None, G=G)
    e1.record(); torch.cuda.synchronize()
    print(json.dumps({"N": N, "rows": B, "ms": round(e0.elapsed_time(e1) / 30, 4),
                      "checksum": float(out[0].double().abs().sum())}), flush=True)
