"""Development: accuracy of the long-context forward (two rows of N = 2^20 inside B = 16) against the C oracle, rows
recomputed, bitwise repeatability.  BKCORE_DEV_LIB=path selects another build of the library."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "project-resnet-bk-an-o-n-language-model-architecture_amd"))
if os.environ.get("BKCORE_DEV_LIB"):
    from bkcore import _lib
    _lib.LIB_PATH = os.path.abspath(os.environ["BKCORE_DEV_LIB"])
import bkcore
from bkcore import bk_core
from oracle import bk_oracle_c as C
B, N = 16, 1 << 20
g = torch.Generator(device="cuda").manual_seed(2020)
a = -2.0 + 0.5 * torch.randn(B, N, generator=g, device="cuda")
b = torch.ones(B, N - 1, device="cuda")
z = torch.tensor(1j, dtype=torch.complex64, device="cuda")
bk_core._DEBUG["keep_ws"] = True
_, G = bkcore.bk_forward(a, b, b, z)
torch.cuda.synchronize()
redone = bk_core.rows_recomputed()
_, G2 = bkcore.bk_forward(a, b, b, z)
rows = [0, 15]
ref = C.forward_f32(a[rows].cpu().numpy(), b[rows].cpu().numpy(), b[rows].cpu().numpy(), 1j)[1]
err = float(np.max(np.abs(G[rows].cpu().numpy() - ref) / np.abs(ref)))
print(f"check_c3: err {err:.2e} rows_recomputed {redone} repeatable {bool(torch.equal(torch.view_as_real(G), torch.view_as_real(G2)))}", flush=True)
