"""Development: one case of tools/dev_fuzz.py again, with detail of the reference-mode backward at its worst element:
the kernel, the C oracle (double) and the reference's own arithmetic (complex64 torch ops, bk_core.py:340-377) on the
same G.  usage: dev_bwd_case.py cases seed want"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "project-resnet-bk-an-o-n-language-model-architecture_amd"))
import bkcore
from oracle import bk_oracle_c as C
from oracle import bk_oracle_torch as O
cases, seed, want = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.RandomState(seed)
for it in range(cases):
    kind = rng.randint(0, 4)
    if kind == 0:
        B, N = int(rng.randint(1, 40)), int(rng.randint(1, 300))
    elif kind == 1:
        B, N = int(rng.randint(1, 1600)), int(rng.randint(4000, 40000))
    elif kind == 2:
        B, N = int(rng.randint(1, 24)), int(rng.randint(100000, 400000))
    else:
        B, N = int(rng.randint(100, 800)), int(rng.randint(8000, 70000))
    while B * N > (1 << 25):
        B = max(1, B // 2)
    dt = [torch.float32, torch.float32, torch.bfloat16, torch.float16, torch.float64][rng.randint(0, 5)]
    exact = bool(rng.randint(0, 2))
    varying = bool(rng.randint(0, 2))
    z = complex(float(rng.uniform(-0.5, 0.5)), float(rng.uniform(0.7, 1.5)))
    gseed = int(rng.randint(0, 1 << 30))
    rows_pick = rng.randint(0, B, size=3).tolist()
    if it != want:
        continue
    g = torch.Generator().manual_seed(gseed)
    a = (-2.0 + 0.5 * torch.randn(B, N, generator=g)).to(dt)
    nb = max(N - 1, 0)
    amp = 0.1 if varying else 0.0
    b = (1.0 + amp * torch.randn(B, nb, generator=g)).to(dt)
    c = (1.0 + amp * torch.randn(B, nb, generator=g)).to(dt)
    gf = torch.randn(B, N, 2, generator=g)
    print(f"case {it}: B={B} N={N} {dt} exact={exact} varying={varying} z={z}")
    _, G = bkcore.bk_forward(a.cuda(), b.cuda(), c.cuda(), z, phi_exact=exact)
    grad = bkcore.bk_backward_reference(G, gf.cuda(), None, 0.5).cpu().numpy()
    Gc = G.cpu()
    gref = C.backward_ref(Gc.numpy(), gf.numpy(), None, 0.5)
    gtorch = O.backward_reference(Gc, gf, None, 0.5).numpy()
    for name, ref in (("C oracle", gref), ("torch oracle", gtorch)):
        e = np.abs(grad - ref) / (np.abs(ref) + 1e-3 * np.abs(ref).max())
        i = np.unravel_index(int(e.argmax()), e.shape)
        Gi = Gc.numpy()[i]
        print(f"vs {name}: worst {e.max():.2e} at {i}: kernel {grad[i]:.6e} ref {ref[i]:.6e} max|ref| {np.abs(ref).max():.3e} "
              f"G {Gi:.6e} |G|^2 {abs(Gi)**2:.4e} g ({gf[i][0]:.4f},{gf[i][1]:.4f})")
    e2 = np.abs(gref - gtorch) / (np.abs(gref) + 1e-3 * np.abs(gref).max())
    print(f"C oracle vs torch oracle: worst {e2.max():.2e}")
