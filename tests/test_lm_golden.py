"""The small ResNet-BK language model against the reference's classic ``LanguageModel`` (src/models/resnet_bk.py:350-483,
use_hybrid_attention=False, one expert) at the dimensions of configs/phase7_small.yaml: loss, norm of every parameter's
gradient, a slice of the logits and the first block's resolvent diagonal, recorded by oracle/gen_golden_lm.py from the
imported reference for the seeded weights and batch of tests/lm_weights.py.

CPU: the model with the CPU oracle plugged in as the operator (host logic + oracle against the reference).
GPU: the model with the HIP operator.
Tolerances (float32 model, bfloat16-rounded operator inputs as in the reference, resnet_bk.py:163-170): loss 1e-4
relative; gradient norms 2e-3 relative (5e-3 on the GPU, whose matrix products accumulate in a different order);
logits 5e-3 absolute (values of order 1); resolvent diagonal 1e-4 relative to its maximum."""
import os

import numpy as np
import pytest
import torch

import lm_weights as W
from test_lm_step import _OracleOp

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lm_golden.npz")


def _model(device):
    from bkcore.lm import SmallResNetBKLM
    m = SmallResNetBKLM(ffn_mult=2, ffn_activation="relu", **W.DIMS)
    state = {k: torch.from_numpy(np.asarray(v)) for k, v in W.synthetic_state(**W.DIMS).items()}
    missing, unexpected = m.load_state_dict(state, strict=True)
    assert not missing and not unexpected
    return m.to(device).eval()


def _check(m, device, gtol):
    d = np.load(GOLD)
    assert int(d["__seed__"]) == W.SEED
    tok, tgt = W.synthetic_batch(vocab_size=W.DIMS["vocab_size"], n_seq=W.DIMS["n_seq"])
    tok, tgt = torch.from_numpy(tok).to(device), torch.from_numpy(tgt).to(device)
    logits = m(tok)
    loss = torch.nn.functional.cross_entropy(logits.reshape(-1, logits.shape[-1]), tgt.reshape(-1))
    loss.backward()
    assert abs(loss.item() - float(d["loss"])) <= 1e-4 * float(d["loss"])
    np.testing.assert_allclose(logits[:, ::16, :64].detach().cpu().numpy(), d["logits_slice"], atol=5e-3, rtol=0)
    g0 = m.blocks[0].bk_layer.last_g_ii.detach().cpu().numpy()[..., 0]
    assert np.abs(g0 - d["g_ii_block0"]).max() <= 1e-4 * np.abs(d["g_ii_block0"]).max()
    params = dict(m.named_parameters())
    worst = ("", 0.0)
    for name in [str(x) for x in d["__param_names__"]]:
        ref = float(d["gradnorm/" + name])
        g = params[name].grad
        if not bool(d["hasgrad/" + name]):
            assert g is None or float(g.norm()) == 0.0, name     # z receives no gradient (bk_core.py:385)
            continue
        got = float(g.double().norm())
        err = abs(got - ref) / max(ref, 1e-12)
        if err > worst[1]:
            worst = (name, err)
    assert worst[1] <= gtol, worst


def test_small_lm_with_oracle_operator_matches_reference_model():
    m = _model("cpu")
    m.set_bk_core(_OracleOp.apply)
    _check(m, "cpu", 2e-3)


@pytest.mark.gpu
def test_small_lm_on_gpu_matches_reference_model():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    _check(_model("cuda"), "cuda", 5e-3)


@pytest.mark.gpu
def test_small_lm_fused_layer_on_gpu_matches_reference_model():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from bkcore.lm import SmallResNetBKLM
    m = SmallResNetBKLM(ffn_mult=2, ffn_activation="relu", fused=True, **W.DIMS)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in W.synthetic_state(**W.DIMS).items()})
    _check(m.cuda().eval(), "cuda", 5e-3)
