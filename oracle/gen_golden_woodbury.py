"""Generate tests/golden/woodbury_golden.npz by running the REFERENCE Birman-Schwinger core (build container only):

    python oracle/gen_golden_woodbury.py [/root/reference]

Calls ``BirmanSchwingerCore.compute_semiseparable_resolvent(v, z)`` (src/models/birman_schwinger_core.py:463-608) with
seeded low-rank factors U, V written into its ``semiseparable`` member, and records v, z, U, V, the tridiagonal
resolvent diagonal (same call with the low-rank part switched off) and the Woodbury-corrected result.  Only inputs and
outputs are stored -- no reference source."""
import os
import sys

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

import numpy as np
import torch

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "woodbury_golden.npz")


def main():
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(REF)
    import warnings
    warnings.filterwarnings("ignore")
    from src.models.birman_schwinger_core import BirmanSchwingerCore
    os.chdir(cwd)
    out, names = {}, []
    # (name, B, N, amplitude of U and V, z)
    cases = [("n64", 3, 64, 0.3, 1j), ("n200", 2, 200, 0.2, 0.3 + 0.8j), ("n1024", 2, 1024, 0.1, 1j),
             ("n17_strong", 4, 17, 1.0, 0.5j)]
    for idx, (name, B, N, amp, z) in enumerate(cases):
        torch.manual_seed(300 + idx)
        core = BirmanSchwingerCore(n_seq=N, use_semiseparable=True)
        r = core.semiseparable.rank
        U = amp * torch.randn(N, r, dtype=torch.float64)
        V = amp * torch.randn(N, r, dtype=torch.float64)
        with torch.no_grad():                       # U, V are nn.Parameters of the reference's SemiseparableMatrix
            core.semiseparable.U.copy_(U.to(core.semiseparable.U.dtype))
            core.semiseparable.V.copy_(V.to(core.semiseparable.V.dtype))
        v = 0.5 * torch.randn(B, N)
        with torch.no_grad():
            G_corr = core.compute_semiseparable_resolvent(v, z)
            assert core.last_semiseparable_active, core.last_semiseparable_reason
            core.use_semiseparable = False
            G_tri = core.compute_semiseparable_resolvent(v, z)
        names.append(name)
        out[f"{name}/v"] = v.numpy()
        out[f"{name}/z"] = np.array([complex(z).real, complex(z).imag])
        out[f"{name}/U"] = core.semiseparable.U.detach().double().numpy()
        out[f"{name}/V"] = core.semiseparable.V.detach().double().numpy()
        out[f"{name}/G_tridiag"] = G_tri.numpy()
        out[f"{name}/G_corrected"] = G_corr.numpy()
        print(f"{name:12s} B={B} N={N} rank={r} dtype={G_corr.dtype} max |corr| = "
              f"{float((G_corr - G_tri).abs().max()):.3e}", flush=True)
    out["__names__"] = np.array(names)
    np.savez_compressed(OUT, **out)
    print("wrote", os.path.normpath(OUT), os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
