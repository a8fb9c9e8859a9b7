"""Generate tests/golden/bkcore_golden_long.npz and tests/golden/cuda_bk_core_golden.npz by running the REFERENCE.

Run in the build container only (the reference checkout is not available on the GPU box; takes a few minutes, the
reference's sequential path costs about 0.55 ms per position whatever the batch):

    python oracle/gen_golden_long.py [/root/reference]

1. The headline sequence length, N = 32768, B = 2, two cases -- the length at which the in-recurrence soft clamp of
   src/models/bk_core.py:60-75,127-136 becomes a 1e-4 effect:
     ``lap``      a ~ N(-2, 0.5^2), b = c = 1, z = 1j           (the distribution of tests/test_bk_core.py:53-56)
     ``varying``  the same diagonal, b, c = 1 + 0.1 randn, z = 1j (position-varying b c: the reference's phi index
                  convention, bk_core.py:165-171)
   Recorded: ``G_raw`` = vmapped_get_diag(a, b, c, z) (bk_core.py:214-216); ``G`` of
   BKCoreFunction.apply(a, b, c, z, False, False) (bk_core.py:239-300; stored only if it differs from G_raw) with the
   check that features == view_as_real(G); BKCoreFunction.backward (bk_core.py:303-385) at GRAD_BLEND = 0.5 for a
   cotangent on the features only, on both outputs, and on G_ii only; the true autograd gradients (a, b, c) through
   vmapped_get_diag in float64.
   Inputs and cotangents are seeded; only a checksum of them is stored and the tests regenerate them (same torch
   build in the container and on the GPU box) and verify the checksum.
2. ``CUDAOptimizedBKCore(n, use_cuda_kernels=False).forward(v)`` (src/models/cuda_bk_core.py:305-413: un-regularised,
   un-scaled recurrence in complex128, Re/Im clamped to +-10) for n in {16, 96}.  That fallback indexes the
   ``expand()``ed off-diagonals by position along the BATCH dimension (cuda_bk_core.py:317,333), so it only runs with
   B == n - 1 rows (any other batch size raises a shape error); the off-diagonals are all ones, which makes the
   numbers right all the same.  The fixtures therefore have n - 1 rows.  (``--only-a5`` regenerates this file alone.)

Only inputs and outputs are stored -- no reference source.
"""
import os
import sys
import time

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

import numpy as np
import torch

REF = next((x for x in sys.argv[1:] if not x.startswith("--")), "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT_LONG = os.path.join(HERE, "..", "tests", "golden", "bkcore_golden_long.npz")
OUT_A5 = os.path.join(HERE, "..", "tests", "golden", "cuda_bk_core_golden.npz")

N_LONG = 32768
B_LONG = 2


def long_inputs(name, n=N_LONG, B=B_LONG):
    """Seeded inputs and cotangents of a long case; the tests call this too."""
    seed = {"lap": 32768001, "varying": 32768002}[name]
    g = torch.Generator().manual_seed(seed)
    a = -2.0 + 0.5 * torch.randn(B, n, generator=g)
    if name == "lap":
        b = torch.ones(B, n - 1)
        c = torch.ones(B, n - 1)
    else:
        b = 1.0 + 0.1 * torch.randn(B, n - 1, generator=g)
        c = 1.0 + 0.1 * torch.randn(B, n - 1, generator=g)
    gf = torch.randn(B, n, 2, generator=g)
    gG = torch.complex(torch.randn(B, n, generator=g), torch.randn(B, n, generator=g))
    return a, b, c, 1j, gf, gG


def checksum(*tensors):
    """float64 sums of x and x * position weight: catches a generator that produced different numbers."""
    out = []
    for t in tensors:
        x = torch.view_as_real(t).double().flatten() if t.is_complex() else t.double().flatten()
        w = torch.arange(x.numel(), dtype=torch.float64) % 1009.0
        out += [float(x.sum()), float((x * w).sum())]
    return np.array(out, dtype=np.float64)


def main():
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(REF)
    import warnings
    warnings.filterwarnings("ignore")
    from src.models.bk_core import BKCoreFunction, vmapped_get_diag
    from src.models.cuda_bk_core import CUDAOptimizedBKCore
    os.chdir(cwd)

    out = {}
    for name in (() if "--only-a5" in sys.argv else ("lap", "varying")):
        t0 = time.time()
        a, b, c, zc, gf, gG = long_inputs(name)
        z = torch.tensor(zc, dtype=torch.complex64)
        out[f"{name}/checksum"] = checksum(a, b, c, gf, gG)
        out[f"{name}/z"] = np.array([zc.real, zc.imag], dtype=np.float64)
        G_raw = vmapped_get_diag(a, b, c, z)
        out[f"{name}/G_raw"] = G_raw.numpy()
        print(f"{name}: get-diag {time.time() - t0:.0f} s", flush=True)

        BKCoreFunction.GRAD_BLEND = 0.5
        ar = a.clone().requires_grad_(True)
        feats, G = BKCoreFunction.apply(ar, b, c, z, False, False)
        assert torch.equal(feats.detach(), torch.view_as_real(G.detach()).float())
        if not torch.equal(G.detach(), G_raw):
            out[f"{name}/G"] = G.detach().numpy()
        lf = (feats * gf).sum()
        lG = (G.real * gG.real + G.imag * gG.imag).sum()
        for tag, loss in (("gG0", lf), ("gG1", lf + lG), ("gGonly", lG)):
            (gr,) = torch.autograd.grad(loss, ar, retain_graph=True)
            out[f"{name}/grad_ref_a0.5_{tag}"] = gr.numpy()
        print(f"{name}: operator forward + reference backward {time.time() - t0:.0f} s", flush=True)

        a64 = a.double().requires_grad_(True)
        b64 = b.double().requires_grad_(True)
        c64 = c.double().requires_grad_(True)
        G64 = vmapped_get_diag(a64, b64, c64, z)
        loss = (G64.real * gf[..., 0] + G64.imag * gf[..., 1]).sum()
        grads = torch.autograd.grad(loss, (a64, b64, c64), allow_unused=True)
        for nm, gr, ref in zip("abc", grads, (a64, b64, c64)):
            out[f"{name}/grad_true_{nm}"] = (torch.zeros_like(ref) if gr is None else gr).float().numpy()
        print(f"{name}: autograd through get-diag {time.time() - t0:.0f} s", flush=True)
    if out:
        out["__torch_version__"] = np.array(torch.__version__)
        np.savez_compressed(OUT_LONG, **out)
        print("wrote", os.path.normpath(OUT_LONG), os.path.getsize(OUT_LONG), "bytes")

    a5 = {}
    for n in (16, 96):
        g = torch.Generator().manual_seed(500 + n)
        v = 0.5 * torch.randn(n - 1, n, generator=g)
        mod = CUDAOptimizedBKCore(n, use_cuda_kernels=False)
        a5[f"n{n}/v"] = v.numpy()
        a5[f"n{n}/features"] = mod(v).detach().numpy()
    np.savez_compressed(OUT_A5, **a5)
    print("wrote", os.path.normpath(OUT_A5), os.path.getsize(OUT_A5), "bytes")


if __name__ == "__main__":
    main()
