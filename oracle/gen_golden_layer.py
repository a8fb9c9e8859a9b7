"""Generate tests/golden/bklayer_golden.npz by running the REFERENCE layer (build container only):

    python oracle/gen_golden_layer.py [/root/reference]

Imports ``MoEResNetBKLayer`` from the reference checkout read-only, removes its FFN branch (``layer.ffn = None``: the
layer then returns ``bk_scale * output_proj(clamp(features))``, src/models/resnet_bk.py:204-219) and takes the
sequential operator path (``BKCoreFunction.USE_PARALLEL_SCAN = False``, as for the operator's own golden vectors).
Records inputs, the BK-Core related parameters / buffers, the output and the gradients of x and of every parameter
that receives one, for a few seeded cases.  Only inputs and outputs are stored -- no reference source."""
import os
import sys

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

import numpy as np
import torch

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "bklayer_golden.npz")
KEEP = ("v_proj.weight", "v_proj.bias", "output_proj.weight", "output_proj.bias", "bk_scale", "gamma", "epsilon_param",
        "h0_diag_base", "h0_sub_base", "h0_super_base")


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
    # (name, B, N, D, scale of v_proj weights, off-diagonal buffers)
    cases = [("identity_init", 2, 32, 16, 1.0, None),
             ("strong_potential", 3, 64, 32, 3000.0, None),
             ("laplacian_buffers", 2, 48, 16, 3000.0, (-2.0, 1.0, 1.0)),
             ("saturating", 2, 40, 8, 60000.0, (-2.0, 1.0, 0.9))]
    for idx, (name, B, N, D, wscale, bufs) in enumerate(cases):
        torch.manual_seed(100 + idx)
        layer = MoEResNetBKLayer(ResNetBKConfig(d_model=D, n_seq=N, num_experts=1))
        layer.ffn = None
        with torch.no_grad():
            layer.v_proj.weight.mul_(wscale)
            layer.output_proj.bias.add_(0.1)
            layer.bk_scale.mul_(1.0 + 0.1 * torch.randn(D))
            if bufs is not None:
                layer.h0_diag_base.fill_(bufs[0])
                layer.h0_super_base.fill_(bufs[1])
                layer.h0_sub_base.fill_(bufs[2])
        x = torch.randn(B, N, D).requires_grad_(True)
        w = torch.randn(B, N, D)
        y = layer(x)
        (y * w).sum().backward()
        names.append(name)
        out[f"{name}/x"] = x.detach().numpy()
        out[f"{name}/w"] = w.numpy()
        out[f"{name}/y"] = y.detach().numpy()
        out[f"{name}/grad_x"] = x.grad.numpy()
        sd = layer.state_dict()
        for k in KEEP:
            out[f"{name}/param/{k}"] = sd[k].detach().numpy()
        for k, p in layer.named_parameters():
            if k in KEEP and p.grad is not None:
                out[f"{name}/grad/{k}"] = p.grad.numpy()
        out[f"{name}/g_ii"] = layer.last_g_ii.detach().numpy()
        print(f"{name:20s} B={B} N={N} D={D} |y| max {float(y.abs().max()):.4f} grads: "
              f"{sorted(k for k, p in layer.named_parameters() if k in KEEP and p.grad is not None)}", flush=True)
    out["__names__"] = np.array(names)
    out["__torch_version__"] = np.array(torch.__version__)
    np.savez_compressed(OUT, **out)
    print("wrote", os.path.normpath(OUT), os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
