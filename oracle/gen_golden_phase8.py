"""Generate tests/golden/bkphase8_golden.npz by running the REFERENCE module whose Green-function step calls the operator
with learned, position-varying off-diagonals (build container only):

    python oracle/gen_golden_phase8.py [/root/reference]

``BKCoreHyperbolicIntegration.compute_green_function`` (src/models/phase8/bk_core_hyperbolic.py:367-407): three Linear
projections of the bfloat16 input, averaged over the model dimension, give he_diag, h0_super and h0_sub; the operator is
called as ``BKCoreFunction.apply(he_diag, h0_super, h0_sub, z, False)``.  Run here on the sequential operator path
(``BKCoreFunction.USE_PARALLEL_SCAN = False``: the caller leaves that flag at its default, and the default path is not
the resolvent) and under ``torch.autocast('cpu', bfloat16)`` (the module asks for CUDA autocast, which a CPU ignores;
without autocast a bfloat16 input meets float32 weights).  Recorded per seeded case: x, the six projection tensors, z,
the three operands as the module formed them, G_ii, the features, a cotangent on the features and the gradients that
reach the projections -- in reference mode only he_diag's: the operator returns none for the off-diagonals
(bk_core.py:384-385).  Only inputs and outputs are stored -- no reference source."""
import os
import sys

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

import numpy as np
import torch

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "bkphase8_golden.npz")
PARAMS = ("he_diag_proj.weight", "he_diag_proj.bias", "h0_super_proj.weight", "h0_super_proj.bias",
          "h0_sub_proj.weight", "h0_sub_proj.bias")


def main():
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(REF)
    import warnings
    warnings.filterwarnings("ignore")
    from src.models.phase8.bk_core_hyperbolic import BKCoreHyperbolicIntegration, BKCoreHyperbolicConfig
    from src.models.bk_core import BKCoreFunction
    import src.models.bk_core as ref_bk
    os.chdir(cwd)
    BKCoreFunction.USE_PARALLEL_SCAN = False
    BKCoreFunction.USE_TRITON = False

    out, names = {}, []
    # (name, B, N, D, z, he_diag bias, he_diag weight scale, off-diagonal bias, off-diagonal weight scale)
    cases = [("identity_init", 2, 48, 16, (0.0, 0.1), None, None, None, None),
             ("laplacian_like", 3, 96, 32, (0.0, 1.0), -2.0, 0.5, 1.0, 0.1),
             ("varying_offdiagonals", 2, 257, 16, (0.3, 0.9), -2.0, 0.4, 1.0, 0.3)]
    for idx, (name, B, N, D, z, bias_a, ws_a, bias_o, ws_o) in enumerate(cases):
        torch.manual_seed(800 + idx)
        m = BKCoreHyperbolicIntegration(BKCoreHyperbolicConfig(d_model=D, z_real=z[0], z_imag=z[1],
                                                               use_green_function_cache=False))
        with torch.no_grad():
            if bias_a is not None:
                m.he_diag_proj.bias.fill_(bias_a)
                m.he_diag_proj.weight.normal_(0.0, ws_a)
                for proj in (m.h0_super_proj, m.h0_sub_proj):
                    proj.bias.fill_(bias_o)
                    proj.weight.normal_(0.0, ws_o)
        x = torch.randn(B, N, D)
        w = torch.randn(B, N, 2)
        seen = {}
        orig = BKCoreFunction.apply

        def spy(he, sup, sub, zz, *rest):
            seen["ops"] = (he.detach(), sup.detach(), sub.detach(), zz.detach(), rest)
            return orig(he, sup, sub, zz, *rest)
        BKCoreFunction.apply = staticmethod(spy)
        try:
            with torch.autocast("cpu", dtype=torch.bfloat16):
                G, feats = m.compute_green_function(x)
        finally:
            BKCoreFunction.apply = orig
        assert "ops" in seen, "the module did not reach the operator"
        (feats * w).sum().backward()
        he, sup, sub, zz, rest = seen["ops"]
        names.append(name)
        out[f"{name}/x"] = x.numpy()
        out[f"{name}/w"] = w.numpy()
        out[f"{name}/z"] = np.array([float(zz.real), float(zz.imag)])
        out[f"{name}/he_diag"] = he.float().numpy()          # (float32 holds bfloat16 values exactly)
        out[f"{name}/h0_super"] = sup.float().numpy()
        out[f"{name}/h0_sub"] = sub.float().numpy()
        out[f"{name}/operand_dtype"] = np.array(str(he.dtype))
        out[f"{name}/g_ii"] = G.detach().numpy()
        out[f"{name}/features"] = feats.detach().numpy()
        sd = m.state_dict()
        params = dict(m.named_parameters())
        for k in PARAMS:
            out[f"{name}/param/{k}"] = sd[k].detach().numpy()
            gk = params[k].grad
            out[f"{name}/grad/{k}"] = np.zeros_like(sd[k].numpy()) if gk is None else gk.float().numpy()
            out[f"{name}/grad_is_none/{k}"] = np.array(gk is None)
        print(f"{name:22s} B={B} N={N} D={D} operands {he.dtype} rest={rest} |G| max {float(G.abs().max()):.3f} "
              f"|grad he_diag_proj.weight| {float(params['he_diag_proj.weight'].grad.abs().max()):.4f} "
              f"off-diagonal grads None: {params['h0_super_proj.weight'].grad is None}", flush=True)
    out["__names__"] = np.array(names)
    out["__torch_version__"] = np.array(torch.__version__)
    np.savez_compressed(OUT, **out)
    print("wrote", os.path.normpath(OUT), os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
