"""Development: for one (N, z) of tools/dev_ana_err_vs_n.py, do the float32 forward and the oracle sit on different
sides of the clip at |G| = 50 anywhere in a row?"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "project-resnet-bk-an-o-n-language-model-architecture_amd"))
import bkcore
from oracle import bk_oracle_c as C
N = int(sys.argv[1]); z = complex(sys.argv[2])
B = 2
g = torch.Generator().manual_seed(N)
a = -2.0 + 0.5 * torch.randn(B, N, generator=g)
b = 1.0 + 0.1 * torch.randn(B, N - 1, generator=g)
c = 1.0 + 0.1 * torch.randn(B, N - 1, generator=g)
_, G = bkcore.bk_forward(a.cuda(), b.cuda(), c.cuda(), z)
ref = C.forward_f32(a.numpy(), b.numpy(), c.numpy(), z)[1]
Gg, Go = np.abs(G.cpu().numpy().astype(np.complex128)), np.abs(ref.astype(np.complex128))
print("forward err", float(np.max(np.abs(G.cpu().numpy() - ref) / np.abs(ref))), "max |G|", Go.max())
for r in range(B):
    flips = np.where((np.abs(Gg[r] - 50.0) < 2e-5) != (np.abs(Go[r] - 50.0) < 2e-5))[0]
    print(f"row {r}: elements at the clip {int((np.abs(Go[r] - 50.0) < 2e-5).sum())}, on different sides: {flips[:8].tolist()}",
          [(float(Gg[r][j]), float(Go[r][j])) for j in flips[:3]])
