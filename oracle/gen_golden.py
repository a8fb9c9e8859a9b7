"""Generate tests/golden/bkcore_golden.npz by running the REFERENCE implementation.

Run in the build container only (the reference checkout is not available on the GPU box):

    python oracle/gen_golden.py [/root/reference]

It imports ``src.models.bk_core`` from the reference checkout read-only
(PYTHONDONTWRITEBYTECODE is forced so nothing is written there) and records, for
a list of small seeded cases:

  * ``G``  = vmapped_get_diag(a, b, c, z)                      (bk_core.py:214-216)
  * ``features, G`` = BKCoreFunction.apply(a, b, c, z, False, False)   (sequential path, bk_core.py:239-300)
  * ``grad_he_diag`` from BKCoreFunction.backward for GRAD_BLEND in {0, 0.5, 1}, with a cotangent on the
    features only, on both outputs, and on the complex output only   (bk_core.py:303-385)
  * true autograd gradients (a, b, c) through vmapped_get_diag in float64  ("analytic" mode oracle)
  * the dense-inverse diagonal for small N (what tests/test_bk_core.py:47-77 compares with)

Only inputs and outputs are stored -- no reference source.
"""
import os
import sys

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

import numpy as np
import torch

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "bkcore_golden.npz")


def build_cases():
    cases = {}

    def add(name, a, b, c, z, z_as_tensor=True, in_dtype=torch.float32):
        cases[name] = dict(a=a.to(in_dtype), b=b.to(in_dtype), c=c.to(in_dtype), z=complex(z),
                           z_as_tensor=z_as_tensor)

    # known-answer: N=4 Laplacian, tests/test_bk_core.py:21-45
    add("kat4", torch.full((1, 4), -2.0), torch.ones(1, 3), torch.ones(1, 3), 1j)

    # seeded Laplacian-like cases (distribution of tests/test_bk_core.py:53-56,86-89)
    for n in (1, 2, 3, 8, 64, 257):
        for B in (1, 8):
            g = torch.Generator().manual_seed(1000 + 17 * n + B)
            a = -2.0 + 0.5 * torch.randn(B, n, generator=g)
            add(f"lap_n{n}_b{B}", a, torch.ones(B, max(n - 1, 0)), torch.ones(B, max(n - 1, 0)), 1j)
    g = torch.Generator().manual_seed(1234)
    add("lap_n1024_b8", -2.0 + 0.5 * torch.randn(8, 1024, generator=g), torch.ones(8, 1023), torch.ones(8, 1023), 1j)

    # diagonal-only, model-like (resnet_bk.py:124-126,155-161): b = c = 0, a = 1 + v, z ~ 1.3133j
    g = torch.Generator().manual_seed(7)
    v = 5.0 * torch.tanh(torch.randn(4, 96, generator=g) / 5.0)
    add("diag_only", 1.0 + v, torch.zeros(4, 95), torch.zeros(4, 95), 1.3133j)
    add("diag_only_bf16", (1.0 + v).to(torch.bfloat16).float(), torch.zeros(4, 95), torch.zeros(4, 95), 1.3133j)

    # learned-off-diagonal style: b != c, position-varying (pins the phi index convention)
    g = torch.Generator().manual_seed(11)
    add("varying_bc", -2.0 + 0.5 * torch.randn(4, 128, generator=g),
        1.0 + 0.1 * torch.randn(4, 127, generator=g), 1.0 + 0.1 * torch.randn(4, 127, generator=g), 1j)
    g = torch.Generator().manual_seed(12)
    add("varying_bc_small", -2.0 + 0.5 * torch.randn(2, 8, generator=g),
        1.0 + 0.3 * torch.randn(2, 7, generator=g), 0.8 + 0.3 * torch.randn(2, 7, generator=g), 0.2 + 0.9j)

    # extremes of tests/test_bk_core.py:111-128
    add("extreme_large", torch.full((1, 16), 100.0), torch.ones(1, 15), torch.ones(1, 15), 1j)
    add("extreme_small", torch.full((1, 16), -0.01), torch.ones(1, 15), torch.ones(1, 15), 1j)
    # out-of-range inputs: exercised through BKCoreFunction (clamps) only
    g = torch.Generator().manual_seed(13)
    a = 60.0 * torch.randn(2, 32, generator=g)
    add("clamped_inputs", a, 8.0 * torch.randn(2, 31, generator=g), 8.0 * torch.randn(2, 31, generator=g), 0.5 + 1j)
    # NaN / Inf poisoned diagonal and off-diagonals
    g = torch.Generator().manual_seed(14)
    a = -2.0 + 0.5 * torch.randn(2, 24, generator=g)
    b = torch.ones(2, 23)
    c = torch.ones(2, 23)
    a[0, 3] = float("nan"); a[1, 7] = float("inf"); a[1, 9] = float("-inf")
    b[0, 5] = float("nan"); c[1, 2] = float("inf")
    add("poisoned", a, b, c, 1j)
    # NaN only in the off-diagonals: the reference sanitiser does NOT trigger (it inspects he_diag only)
    a2 = -2.0 + 0.5 * torch.randn(1, 12, generator=g)
    b2 = torch.ones(1, 11); b2[0, 4] = float("nan")
    add("nan_offdiag_only", a2, b2, torch.ones(1, 11), 1j)
    # z given as a python complex instead of a tensor, real part non-zero
    g = torch.Generator().manual_seed(15)
    add("z_python_complex", -2.0 + 0.5 * torch.randn(3, 40, generator=g), torch.ones(3, 39), torch.ones(3, 39),
        0.3 + 0.7j, z_as_tensor=False)
    # reference benchmark distribution (scripts/benchmark_bk_triton.py:66-70): ill-conditioned
    g = torch.Generator().manual_seed(16)
    add("bench_randn", torch.randn(4, 64, generator=g), torch.randn(4, 63, generator=g),
        torch.randn(4, 63, generator=g), 0.1 + 0.1j)
    # float64 inputs as the Birman-Schwinger caller passes them (birman_schwinger_core.py:523-540)
    g = torch.Generator().manual_seed(18)
    add("f64_inputs", -2.0 + 0.5 * torch.randn(2, 48, generator=g, dtype=torch.float64),
        torch.ones(2, 47, dtype=torch.float64), torch.ones(2, 47, dtype=torch.float64), 1j, in_dtype=torch.float64)
    # strong growth / decay: large log-magnitudes exercise both soft clamps
    g = torch.Generator().manual_seed(19)
    add("big_diag", 30.0 + 5.0 * torch.randn(2, 200, generator=g), torch.ones(2, 199), torch.ones(2, 199), 1j)
    add("tiny_diag_big_k", 0.05 * torch.randn(2, 64, generator=g), 3.0 + 0.2 * torch.randn(2, 63, generator=g),
        3.0 + 0.2 * torch.randn(2, 63, generator=g), 0.5j)
    return cases


def main():
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(REF)
    import warnings
    warnings.filterwarnings("ignore")
    from src.models.bk_core import BKCoreFunction, vmapped_get_diag
    os.chdir(cwd)

    out = {}
    names = []
    for name, cs in build_cases().items():
        a, b, c, zc = cs["a"], cs["b"], cs["c"], cs["z"]
        z = torch.tensor(zc, dtype=torch.complex64) if cs["z_as_tensor"] else zc
        B, n = a.shape
        names.append(name)
        out[f"{name}/a"] = a.numpy()
        out[f"{name}/b"] = b.numpy()
        out[f"{name}/c"] = c.numpy()
        out[f"{name}/z"] = np.array([zc.real, zc.imag], dtype=np.float64)

        # raw batched get-diag (no clamps) -- only where inputs are finite
        finite = bool(torch.isfinite(a).all() and torch.isfinite(b).all() and torch.isfinite(c).all())
        if finite:
            out[f"{name}/G_raw"] = vmapped_get_diag(a, b, c, z).numpy()

        # operator forward/backward on the sequential path
        g = torch.Generator().manual_seed(4242 + n)
        gf = torch.randn(B, n, 2, generator=g)
        gG = torch.complex(torch.randn(B, n, generator=g), torch.randn(B, n, generator=g))
        out[f"{name}/gfeat"] = gf.numpy()
        out[f"{name}/gG"] = gG.numpy()
        for alpha in (0.0, 0.5, 1.0):
            BKCoreFunction.GRAD_BLEND = alpha
            # cotangent on the features only / on both outputs / on G_ii only (the features unused: what the hybrid
            # attention layer of configs/phase7_small.yaml does with the operator, resnet_bk.py:172-185)
            for tag in ("gG0", "gG1", "gGonly"):
                ar = a.clone().requires_grad_(True)
                feats, G = BKCoreFunction.apply(ar, b, c, z, False, False)
                loss = 0.0
                if tag != "gGonly":
                    loss = loss + (feats * gf).sum()
                if tag != "gG0":
                    loss = loss + (G.real * gG.real + G.imag * gG.imag).sum()
                loss.backward()
                out[f"{name}/grad_ref_a{alpha}_{tag}"] = ar.grad.numpy()
        BKCoreFunction.GRAD_BLEND = 0.5
        out[f"{name}/features"] = feats.detach().numpy()
        out[f"{name}/G"] = G.detach().numpy()

        # true autograd gradients through the sequential path (float64 inputs), finite inputs only
        if finite:
            a64 = a.double().clamp(-100, 100).requires_grad_(True)
            b64 = b.double().clamp(-10, 10).requires_grad_(True)
            c64 = c.double().clamp(-10, 10).requires_grad_(True)
            G64 = vmapped_get_diag(a64, b64, c64, z)
            loss = (G64.real * gf[..., 0] + G64.imag * gf[..., 1]).sum()
            grads = torch.autograd.grad(loss, (a64, b64, c64), allow_unused=True)
            for nm, gr, ref in zip("abc", grads, (a64, b64, c64)):
                out[f"{name}/grad_true_{nm}"] = (torch.zeros_like(ref) if gr is None else gr).numpy()
            if n <= 64:
                T = torch.zeros(B, n, n, dtype=torch.complex128)
                i = torch.arange(n)
                T[:, i, i] = a.double().clamp(-100, 100).to(torch.complex128) - zc
                if n > 1:
                    T[:, i[:-1], i[1:]] = b.double().clamp(-10, 10).to(torch.complex128)
                    T[:, i[1:], i[:-1]] = c.double().clamp(-10, 10).to(torch.complex128)
                out[f"{name}/G_dense"] = torch.diagonal(torch.linalg.inv(T), dim1=-2, dim2=-1).numpy()
        print(f"{name:24s} B={B} N={n} done", flush=True)

    out["__names__"] = np.array(names)
    out["__torch_version__"] = np.array(torch.__version__)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    np.savez_compressed(OUT, **out)
    print("wrote", os.path.normpath(OUT), os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
