"""Development: many calls of the forward in units with the hand-over statistics of a BK_RELAY_DEBUG build
(bash tools/build_variant.sh dbg "-DBK_RELAY_DEBUG=1" first): waits per call, longest wait, give-ups, repeatability."""
import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "project-resnet-bk-an-o-n-language-model-architecture_amd"))
from bkcore import _lib
_lib.LIB_PATH = os.path.abspath(os.path.join(ROOT, "tools/ab/libbkcore_dbg.so"))
import bkcore
from bkcore import bk_core
for B in (1024, 900, 1300):
    N = 32768
    g = torch.Generator(device="cuda").manual_seed(B)
    a = -2.0 + 0.5 * torch.randn(B, N, device="cuda", generator=g)
    b = torch.ones(B, N - 1, device="cuda")
    z = torch.tensor(1j, dtype=torch.complex64, device="cuda")
    _, G0 = bkcore.bk_forward(a, b, b, z, want_features=False); G0 = G0.clone()
    st = bk_core._state_block(a.device, torch.cuda.current_stream().cuda_stream)
    st[13:16] = 0
    gaveup = 0
    for it in range(5000):
        _, G = bkcore.bk_forward(a, b, b, z, want_features=False)
        if it % 500 == 499:
            torch.cuda.synchronize()
            gaveup += int(not torch.equal(torch.view_as_real(G), torch.view_as_real(G0)))
    torch.cuda.synchronize()
    s = st.cpu().tolist()
    print(json.dumps({"B": B, "calls": 5000, "waits": s[15], "max_wait_us": s[14] / 100.0, "mismatches": gaveup, "state12": s[12], "epoch": s[9]}), flush=True)
