"""Development: the in-launch hand-off of rows cut into units (mode 3 of the ratio-form kernel): time and poll
statistics.  usage: BKCORE_RELAY_UNITS=S dev_relay.py [--lib path] [--B rows]"""
import argparse, os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "project-resnet-bk-an-o-n-language-model-architecture_amd"))
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--B", type=int, default=1024)
ap.add_argument("--N", type=int, default=32768)
args = ap.parse_args()
from bkcore import _lib
if args.lib:
    _lib.LIB_PATH = os.path.abspath(args.lib)
import bkcore
from bkcore import bk_core
B, N = args.B, args.N
g = torch.Generator(device="cuda").manual_seed(1234)
a = -2.0 + 0.5 * torch.randn(B, N, device="cuda", generator=g)
b = torch.ones(B, N - 1, device="cuda"); c = torch.ones(B, N - 1, device="cuda")
z = torch.tensor(1j, dtype=torch.complex64, device="cuda")
fn = lambda: bkcore.bk_forward(a, b, c, z, want_features=False)
for _ in range(5): fn()
torch.cuda.synchronize()
st = bk_core._state_block(a.device, torch.cuda.current_stream().cuda_stream)
before = st.cpu().tolist()
it = 50
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(it): fn()
e1.record(); torch.cuda.synchronize()
after = st.cpu().tolist()
print(json.dumps({"units": os.environ.get("BKCORE_RELAY_UNITS"), "lib": args.lib, "B": B, "ms": e0.elapsed_time(e1) / it,
                  "polls_per_call": (after[13] - before[13]) / it, "max_polls": after[14],
                  "waits_per_call": (after[15] - before[15]) / it, "epoch": after[9], "state": after}), flush=True)
