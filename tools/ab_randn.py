"""Development: the reference benchmark's input distribution (randn a, b, c; z = 0.1+0.1j; scripts/benchmark_bk_triton.py:66-70)
at the published shape (host timer + synchronize per call) and at the headline shape (stream events), plus the share of
rows the second launch recomputed.  BKCORE_DEV_LIB=path selects another build of the library."""
import os, sys, time, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "project-resnet-bk-an-o-n-language-model-architecture_amd"))
if os.environ.get("BKCORE_DEV_LIB"):
    from bkcore import _lib
    _lib.LIB_PATH = os.path.abspath(os.environ["BKCORE_DEV_LIB"])
import bkcore

dev = "cuda"
gp = torch.Generator(device=dev).manual_seed(0)
zp = torch.tensor(0.1 + 0.1j, dtype=torch.complex64, device=dev)
out = {}
a, b, c = (torch.randn(16, n, generator=gp, device=dev) for n in (4096, 4095, 4095))
for _ in range(20):
    bkcore.BKCoreFunction.apply(a, b, c, zp, False)
torch.cuda.synchronize()
ts = []
for _ in range(200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bkcore.BKCoreFunction.apply(a, b, c, zp, False)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
out["published_ms_mean"] = sum(ts) / len(ts) * 1e3
out["published_ms_min"] = min(ts) * 1e3
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200):
    bkcore.bk_forward(a, b, c, zp)
e1.record(); torch.cuda.synchronize()
out["published_ms_events"] = e0.elapsed_time(e1) / 200
B, N = int(os.environ.get('AB_ROWS', '1024')), 32768
out['rows'] = B
a, b, c = (torch.randn(B, n, generator=gp, device=dev) for n in (N, N - 1, N - 1))
for _ in range(3):
    bkcore.bk_forward(a, b, c, zp)
e0.record()
for _ in range(20):
    bkcore.bk_forward(a, b, c, zp)
e1.record(); torch.cuda.synchronize()
out["headline_shape_ms"] = e0.elapsed_time(e1) / 20
print(json.dumps(out))
