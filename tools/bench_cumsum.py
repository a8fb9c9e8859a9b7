"""Cumulative sum along dim 1 of (B, L, rank) and along the last dim: library kernel vs torch.cumsum."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "project-resnet-bk-an-o-n-language-model-architecture_amd"))
import torch, bkcore

def timed(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

for shape, dim in (((8, 2048, 32), 1), ((8, 8192, 32), 1), ((64, 8192, 32), 1), ((256, 32768, 32), 1),
                   ((4, 8192, 512), 1), ((4096, 8192), 1), ((32768, 32768), 1)):
    x = torch.randn(*shape, device="cuda")
    t_lib = timed(lambda: bkcore.fused_associative_scan(x, dim=dim))
    t_ref = timed(lambda: torch.cumsum(x, dim))
    print(json.dumps({"shape": shape, "dim": dim, "bkcore_ms": t_lib, "torch_cumsum_ms": t_ref,
                      "GBps": 8 * x.numel() / t_lib / 1e6}))
