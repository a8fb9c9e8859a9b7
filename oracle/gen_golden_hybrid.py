"""Generate tests/golden/bkhybrid_golden.npz by running the REFERENCE layer that configs/phase7_small.yaml selects
(build container only):

    python oracle/gen_golden_hybrid.py [/root/reference]

``MoEResNetBKLayer(use_hybrid_attention=True)`` (src/models/resnet_bk.py:137-202) on the sequential operator path
(``BKCoreFunction.USE_PARALLEL_SCAN = False``).  The layer's own arithmetic runs in float32; only its
``HybridHyperbolicAttention`` submodule is called under ``torch.autocast('cpu', bfloat16)``, without which it does not
run on a CPU (src/models/phase7/hyperbolic_attention.py:327 feeds bfloat16 into a float32 Linear).  In that layer the
operator's features go unused, ``G_ii`` feeds the gate of hybrid_attention.py:117-143, and the operator's backward
receives a cotangent on ``G_ii`` alone.  Recorded per seeded case: x, the cotangent w, the BK-Core parameters and
buffers, ``gate_sensitivity`` and the LayerNorm of the hybrid module, the outputs of its two mixing paths (windowed
hyperbolic attention and AR-SSM, captured by forward hooks -- they are inputs of the gate, not part of this package), the
layer output, ``last_g_ii``, and the gradients of ``v_proj`` and ``gate_sensitivity``.
Only inputs and outputs are stored -- no reference source."""
import os
import sys

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

import numpy as np
import torch

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "bkhybrid_golden.npz")
KEEP = ("v_proj.weight", "v_proj.bias", "output_proj.weight", "output_proj.bias", "bk_scale", "gamma", "epsilon_param",
        "h0_diag_base", "h0_sub_base", "h0_super_base", "hybrid_attn.gate_sensitivity", "hybrid_attn.layer_norm.weight",
        "hybrid_attn.layer_norm.bias")
GRADS = ("v_proj.weight", "v_proj.bias", "hybrid_attn.gate_sensitivity", "hybrid_attn.layer_norm.weight",
         "hybrid_attn.layer_norm.bias")


def main():
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(REF)
    import warnings
    warnings.filterwarnings("ignore")
    from src.models.resnet_bk import MoEResNetBKLayer, ResNetBKConfig
    from src.models.bk_core import BKCoreFunction
    os.chdir(cwd)
    BKCoreFunction.USE_PARALLEL_SCAN = False
    BKCoreFunction.USE_TRITON = False

    out, names = {}, []
    # (name, B, N, D, heads, window, scale of v_proj weights, gate sensitivity, off-diagonal buffers)
    cases = [("identity_init", 2, 64, 32, 4, 16, 3000.0, 1.0, None),
             ("laplacian_buffers", 3, 96, 16, 2, 32, 3000.0, 2.5, (-2.0, 1.0, 1.0)),
             ("ragged_window", 2, 40, 16, 2, 16, 20000.0, 0.7, (-2.0, 1.0, 0.9))]
    for idx, (name, B, N, D, heads, window, wscale, sens, bufs) in enumerate(cases):
        torch.manual_seed(300 + idx)
        layer = MoEResNetBKLayer(ResNetBKConfig(d_model=D, n_seq=N, num_experts=1, num_heads=heads,
                                                use_hybrid_attention=True, hyperbolic_window_size=window))
        assert layer.ffn is None
        with torch.no_grad():
            layer.v_proj.weight.mul_(wscale)
            layer.hybrid_attn.gate_sensitivity.fill_(sens)
            layer.hybrid_attn.layer_norm.weight.mul_(1.0 + 0.1 * torch.randn(D))
            layer.hybrid_attn.layer_norm.bias.add_(0.05 * torch.randn(D))
            if bufs is not None:
                layer.h0_diag_base.fill_(bufs[0])
                layer.h0_super_base.fill_(bufs[1])
                layer.h0_sub_base.fill_(bufs[2])
        hyb = layer.hybrid_attn
        inner = hyb.forward

        def under_autocast(x, g_ii=None, return_diagnostics=True, _inner=inner):
            with torch.autocast("cpu", dtype=torch.bfloat16):
                return _inner(x, g_ii=g_ii, return_diagnostics=return_diagnostics)
        hyb.forward = under_autocast
        seen = {}
        h1 = hyb.local_attn.register_forward_hook(lambda m, i, o: seen.__setitem__("local", o[0].detach()))
        h2 = hyb.global_attn.register_forward_hook(lambda m, i, o: seen.__setitem__("global", o[0].detach()))
        x = torch.randn(B, N, D).requires_grad_(True)
        w = torch.randn(B, N, D)
        y = layer(x)
        (y.float() * w).sum().backward()
        h1.remove(); h2.remove()
        names.append(name)
        out[f"{name}/x"] = x.detach().numpy()
        out[f"{name}/w"] = w.numpy()
        out[f"{name}/y"] = y.detach().float().numpy()
        # (stored as float32, which holds bfloat16 values exactly; the dtype they had is recorded beside them)
        out[f"{name}/local_out"] = seen["local"].float().reshape(B, -1, D)[:, :N].numpy()      # hybrid_attention.py:107-108
        out[f"{name}/global_out"] = seen["global"].float().numpy()
        out[f"{name}/local_dtype"] = np.array(str(seen["local"].dtype))
        out[f"{name}/global_dtype"] = np.array(str(seen["global"].dtype))
        out[f"{name}/g_ii"] = layer.last_g_ii.detach().numpy()
        sd = layer.state_dict()
        for k in KEEP:
            out[f"{name}/param/{k}"] = sd[k].detach().numpy()
        params = dict(layer.named_parameters())
        for k in GRADS:
            out[f"{name}/grad/{k}"] = params[k].grad.float().numpy()
        for k in ("output_proj.weight", "bk_scale", "gamma", "epsilon_param"):
            assert params[k].grad is None or float(params[k].grad.abs().max()) == 0.0, k   # unused / no gradient for z
        print(f"{name:20s} B={B} N={N} D={D} |y| max {float(y.abs().max()):.3f} gate grad "
              f"{float(params['hybrid_attn.gate_sensitivity'].grad):+.4f} |v_proj grad| "
              f"{float(params['v_proj.weight'].grad.abs().max()):.4f}", flush=True)
    out["__names__"] = np.array(names)
    out["__torch_version__"] = np.array(torch.__version__)
    np.savez_compressed(OUT, **out)
    print("wrote", os.path.normpath(OUT), os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
