"""Generate tests/golden/lm_golden.npz by running the REFERENCE classic language model (build container only):

    python oracle/gen_golden_lm.py [/root/reference]

Imports ``LanguageModel`` from the reference checkout read-only (src/models/resnet_bk.py:350-483) with
``use_hybrid_attention=False`` at the dimensions of configs/phase7_small.yaml (d_model 256, 4 layers, n_seq 256, vocab
50257) and a single expert (``num_experts=1``: the feed-forward branch is then one Linear-ReLU-Linear, moe.py:61-69),
fills it with the seeded synthetic weights of tests/lm_weights.py, takes the sequential operator path
(``BKCoreFunction.USE_PARALLEL_SCAN = False``), and records for the seeded batch: the cross-entropy loss, the norm of the
gradient of every parameter, a slice of the logits and the resolvent diagonal of the first block.  Only inputs and
outputs are stored -- no reference source, and no weights (the tests rebuild them from the seed)."""
import os
import sys

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

import numpy as np
import torch

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden", "lm_golden.npz")
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
import lm_weights as W  # noqa: E402


def main():
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(REF)
    import warnings
    warnings.filterwarnings("ignore")
    from src.models.resnet_bk import LanguageModel
    from src.models.config import ResNetBKConfig
    from src.models.bk_core import BKCoreFunction
    os.chdir(cwd)
    BKCoreFunction.USE_PARALLEL_SCAN = False
    BKCoreFunction.USE_TRITON = False
    torch.manual_seed(0)
    cfg = ResNetBKConfig(use_hybrid_attention=False, num_experts=1, top_k=1, dropout_p=0.0, **W.DIMS)
    model = LanguageModel(cfg)
    model.eval()
    state = W.synthetic_state(**W.DIMS)
    ref_state = model.state_dict()
    used = set()
    with torch.no_grad():
        for name, arr in state.items():
            rn = W.to_reference_name(name)
            assert rn in ref_state, rn
            assert tuple(ref_state[rn].shape) == tuple(np.shape(arr)), (rn, ref_state[rn].shape, np.shape(arr))
            ref_state[rn].copy_(torch.from_numpy(np.asarray(arr)))
            used.add(rn)
    left = sorted(k for k in ref_state if k not in used)
    print("reference entries not set from the synthetic state:", left)   # the gate of the single expert, counters
    model.load_state_dict(ref_state)
    tok, tgt = W.synthetic_batch(vocab_size=W.DIMS["vocab_size"], n_seq=W.DIMS["n_seq"])
    tok, tgt = torch.from_numpy(tok), torch.from_numpy(tgt)
    logits = model(tok)
    loss = torch.nn.functional.cross_entropy(logits.reshape(-1, logits.shape[-1]), tgt.reshape(-1))
    loss.backward()
    out = {"loss": np.float64(loss.item()), "logits_slice": logits[:, ::16, :64].detach().numpy(),
           "g_ii_block0": model.blocks[0].bk_layer.last_g_ii.detach().numpy()[..., 0]}
    params = dict(model.named_parameters())
    names = []
    for name in state:
        rn = W.to_reference_name(name)
        if rn in params:
            g = params[rn].grad
            names.append(name)
            out["gradnorm/" + name] = np.float64(0.0 if g is None else g.double().norm().item())
            out["hasgrad/" + name] = np.bool_(g is not None)
    out["__param_names__"] = np.array(names)
    out["__torch_version__"] = np.array(torch.__version__)
    out["__seed__"] = np.int64(W.SEED)
    np.savez_compressed(OUT, **out)
    print(f"loss {loss.item():.6f}; wrote", os.path.normpath(OUT), os.path.getsize(OUT), "bytes")
    for k in names[:14]:
        print(f"  {k:45s} {out['gradnorm/' + k]:.6e}")


if __name__ == "__main__":
    main()
