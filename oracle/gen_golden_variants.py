"""Generate tests/golden/variants_golden.npz from the REFERENCE's other backward wrappers (build container only):

    python oracle/gen_golden_variants.py [/root/reference]

Imports ``batched_compute_gradient`` / ``BatchedAnalyticBKCoreFunction`` / ``MemoryOptimizedBKCoreFunction``
(src/models/batched_gradient.py) and ``BKScanStableFunction`` (src/kernels/bk_scan_stable.py) from the reference checkout
read-only and records, for seeded saved resolvent diagonals G (ordinary, tiny |G| below the 1e-3 floor of |G^2|, large,
one non-finite entry) and cotangents: the gradients their backward passes return.  Only inputs and outputs are stored."""
import os
import sys

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

import numpy as np
import torch

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "variants_golden.npz")


class _Ctx:
    def __init__(self, saved, needs):
        self.saved_tensors = saved
        self.needs_input_grad = needs
        self.z = 1j


def main():
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(REF)
    import warnings
    warnings.filterwarnings("ignore")
    from src.models.batched_gradient import (BatchedAnalyticBKCoreFunction, MemoryOptimizedBKCoreFunction,
                                             batched_compute_gradient)
    from src.kernels.bk_scan_stable import BKScanStableFunction
    os.chdir(cwd)
    g = torch.Generator().manual_seed(77)
    B, N = 5, 203
    mag = torch.exp(torch.empty(B, N).uniform_(-6.0, 3.0, generator=g))            # |G| from 2.5e-3 to 20
    ph = torch.empty(B, N).uniform_(-3.14159, 3.14159, generator=g)
    G = torch.polar(mag, ph).to(torch.complex64)
    G[0, 5] = 0.0
    G[1, 7] = complex(float("inf"), 0.0)
    gf = torch.randn(B, N, 2, generator=g)
    gG = torch.complex(torch.randn(B, N, generator=g), torch.randn(B, N, generator=g)).to(torch.complex64)
    out = {"G": G.numpy(), "gfeat": gf.numpy(), "gG": gG.numpy()}
    for alpha in (0.0, 0.5, 1.0):
        ref = batched_compute_gradient(G, torch.complex(gf[..., 0], gf[..., 1]), alpha)
        out[f"batched_a{alpha}"] = ref.to(torch.float32).numpy()
    # through the classes' own backward (GRAD_BLEND class attribute, chunked variant)
    for cls, key in ((BatchedAnalyticBKCoreFunction, "batched_cls"), (MemoryOptimizedBKCoreFunction, "memopt_cls")):
        cls.GRAD_BLEND = 0.25
        res = cls.backward(_Ctx((G,), (True, False, False, False)), gf)
        cls.GRAD_BLEND = 0.5
        assert res[1] is None and res[2] is None and res[3] is None
        out[key + "_a0.25"] = res[0].numpy()
    a = torch.zeros(B, N)
    b = torch.ones(B, N - 1)
    res = BKScanStableFunction.backward(_Ctx((a, b, b, G), (True, True, True, False, False)), gG)
    out["scan_stable_grad_a"] = res[0].numpy()
    assert float(res[1].abs().max()) == 0.0 and float(res[2].abs().max()) == 0.0
    out["__torch_version__"] = np.array(torch.__version__)
    np.savez_compressed(OUT, **out)
    print("wrote", os.path.normpath(OUT), os.path.getsize(OUT), "bytes")
    for k in sorted(out):
        if not k.startswith("__"):
            v = np.asarray(out[k])
            print(f"  {k:22s} {v.shape} finite {bool(np.isfinite(v.view(np.float32) if v.dtype == np.complex64 else v).all())}")


if __name__ == "__main__":
    main()
