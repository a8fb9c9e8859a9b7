"""Generate tests/golden/bkcomplex_golden.npz by running the REFERENCE's raw resolvent diagonal on complex operands
(build container only):

    python oracle/gen_golden_complex.py [/root/reference]

``get_tridiagonal_inverse_diagonal`` converts every operand to complex128 (src/models/bk_core.py:45-47) and so accepts
complex he_diag, h0_super and h0_sub; ``vmapped_get_diag`` (bk_core.py:214-216) batches it.  Recorded per seeded case:
the operands (complex64, or real where the case keeps an operand real), z and ``vmapped_get_diag(a, b, c, z)``.
Also recorded: that ``BKCoreFunction.apply`` refuses complex operands (its input clamp, bk_core.py:256-258, raises for
them), which is why there is no clamped form, no features tensor and no backward for this case.
Only inputs and outputs are stored -- no reference source."""
import os
import sys

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

import numpy as np
import torch

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "bkcomplex_golden.npz")


def cplx(g, shape, re_mean, re_std, im_std):
    return torch.complex(re_mean + re_std * torch.randn(shape, generator=g), im_std * torch.randn(shape, generator=g))


def cases():
    """name -> (a, b, c, z); seeded."""
    out = {}
    g = torch.Generator().manual_seed(4501)
    out["complex_diagonal"] = (cplx(g, (2, 64), -2.0, 0.5, 0.3), torch.ones(2, 63), torch.ones(2, 63), 1j)
    g = torch.Generator().manual_seed(4502)
    out["complex_offdiagonals"] = (-2.0 + 0.5 * torch.randn(2, 257, generator=g), cplx(g, (2, 256), 1.0, 0.2, 0.2),
                                   cplx(g, (2, 256), 1.0, 0.2, 0.2), 0.3 + 0.9j)
    g = torch.Generator().manual_seed(4503)
    out["all_complex"] = (cplx(g, (2, 1024), -2.0, 0.5, 0.4), cplx(g, (2, 1023), 1.0, 0.1, 0.3),
                          cplx(g, (2, 1023), 1.0, 0.1, 0.3), 0.1 + 1.0j)
    g = torch.Generator().manual_seed(4504)      # nothing contracts: the distribution of scripts/benchmark_bk_triton.py
    out["not_contracting"] = (cplx(g, (3, 512), 0.0, 1.0, 1.0), cplx(g, (3, 511), 0.0, 1.0, 1.0),
                              cplx(g, (3, 511), 0.0, 1.0, 1.0), 0.1 + 0.1j)
    g = torch.Generator().manual_seed(4505)      # three chunks of the complex kernel, ragged end
    out["three_chunks"] = (cplx(g, (1, 4100), -2.0, 0.5, 0.3), cplx(g, (1, 4099), 1.0, 0.1, 0.2),
                           cplx(g, (1, 4099), 1.0, 0.1, 0.2), 1j)
    g = torch.Generator().manual_seed(4506)
    out["n1"] = (cplx(g, (3, 1), -2.0, 0.5, 0.5), torch.zeros(3, 0, dtype=torch.complex64),
                 torch.zeros(3, 0, dtype=torch.complex64), 1j)
    out["n2"] = (cplx(g, (3, 2), -2.0, 0.5, 0.5), cplx(g, (3, 1), 1.0, 0.2, 0.2), cplx(g, (3, 1), 1.0, 0.2, 0.2), 0.5j)
    g = torch.Generator().manual_seed(4507)      # a diagonal entry equal to z: the regularisation's |x| < 1e-12 branch
    a = cplx(g, (1, 33), -2.0, 0.5, 0.3)
    a[0, 7] = 0.25 + 0.75j
    out["diagonal_meets_z"] = (a, cplx(g, (1, 32), 1.0, 0.1, 0.1), cplx(g, (1, 32), 1.0, 0.1, 0.1), 0.25 + 0.75j)
    return out


def main():
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(REF)
    import warnings
    warnings.filterwarnings("ignore")
    from src.models.bk_core import vmapped_get_diag, get_tridiagonal_inverse_diagonal, BKCoreFunction
    os.chdir(cwd)
    BKCoreFunction.USE_TRITON = False
    out, names = {}, []
    for name, (a, b, c, z) in cases().items():
        zt = torch.tensor(z, dtype=torch.complex64)
        with torch.no_grad():
            G = vmapped_get_diag(a, b, c, zt)
            g0 = get_tridiagonal_inverse_diagonal(a[0], b[0], c[0], zt)
        assert G.dtype == torch.complex64 and torch.equal(G[0], g0)
        names.append(name)
        out[f"{name}/a"], out[f"{name}/b"], out[f"{name}/c"] = a.numpy(), b.numpy(), c.numpy()
        out[f"{name}/z"] = np.array([z.real, z.imag])
        out[f"{name}/G"] = G.numpy()
        print(f"{name:22s} a {tuple(a.shape)} {a.dtype}, b {b.dtype}: |G| max {float(G.abs().max()):.4f} "
              f"min {float(G.abs().min()):.3e} clipped {int((G.abs() > 49.999).sum())}", flush=True)
    a, b, c, z = cases()["complex_diagonal"]
    try:
        BKCoreFunction.apply(a, b, c, torch.tensor(z, dtype=torch.complex64), False, False)
        refused = False
    except Exception as e:                                             # noqa: BLE001
        refused = True
        print("BKCoreFunction.apply on complex operands:", type(e).__name__, str(e)[:100])
    out["operator_refuses_complex"] = np.array(refused)
    out["__names__"] = np.array(names)
    out["__torch_version__"] = np.array(torch.__version__)
    np.savez_compressed(OUT, **out)
    print("wrote", os.path.normpath(OUT), os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
