"""The consumer of the operator that configs/phase7_small.yaml selects -- ``MoEResNetBKLayer(use_hybrid_attention=True)``,
src/models/resnet_bk.py:137-202 with the gate of src/models/phase7/hybrid_attention.py:117-143 -- against outputs and
gradients of the reference layer itself (tests/golden/bkhybrid_golden.npz, oracle/gen_golden_hybrid.py).  The fixture
supplies what the reference's two mixing paths produced (they are inputs of the gate and not part of this package);
pinned here: v_proj -> saturation -> bfloat16 operands -> operator -> G_ii -> gate -> mix -> LayerNorm, and backward
from there into v_proj, where the operator's backward gets a cotangent on G_ii and none on the features.

CPU: the CPU oracle plugged in as the operator.  GPU: the HIP operator, plain and fused layer steps.
Tolerances as for the classic layer fixture: the bulk tightly, every element to a bfloat16 step (a last-bit difference
of tanh moves an element of the bfloat16 diagonal by 2^-8 of its size)."""
import os

import numpy as np
import pytest
import torch

from test_layer_golden import _close
from test_lm_step import _OracleOp

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bkhybrid_golden.npz")
BK_KEYS = ("v_proj.weight", "v_proj.bias", "output_proj.weight", "output_proj.bias", "bk_scale", "gamma",
           "epsilon_param", "h0_diag_base", "h0_sub_base", "h0_super_base", "hybrid_attn.gate_sensitivity",
           "hybrid_attn.layer_norm.weight", "hybrid_attn.layer_norm.bias")
GRADS = ("v_proj.weight", "v_proj.bias", "hybrid_attn.gate_sensitivity", "hybrid_attn.layer_norm.weight",
         "hybrid_attn.layer_norm.bias")
DT = {"torch.float32": torch.float32, "torch.bfloat16": torch.bfloat16}


def _run(device, fused=False, oracle=False):
    from bkcore.layers import BKHybridLayer
    z = np.load(GOLD, allow_pickle=False)
    names = [str(n) for n in z["__names__"]]
    assert len(names) >= 3
    for n in names:
        x0 = torch.from_numpy(z[f"{n}/x"])
        B, N, D = x0.shape
        layer = BKHybridLayer(D, N, num_heads=2, window=8, rank=4, fused=fused)
        missing, unexpected = layer.load_state_dict({k: torch.from_numpy(z[f"{n}/param/{k}"]) for k in BK_KEYS},
                                                    strict=False)
        assert not unexpected and all(m.startswith(("local_", "global_")) for m in missing)   # the stand-in paths
        layer = layer.to(device)
        if oracle:
            layer.bk_core = _OracleOp.apply
        x = x0.to(device).requires_grad_(True)
        local = torch.from_numpy(z[f"{n}/local_out"]).to(device).to(DT[str(z[f"{n}/local_dtype"])])
        glob = torch.from_numpy(z[f"{n}/global_out"]).to(device).to(DT[str(z[f"{n}/global_dtype"])])
        _, G_ii = layer.spectral_features(x)
        y = layer.hybrid_attn(local, glob, G_ii)
        (y.float() * torch.from_numpy(z[f"{n}/w"]).to(device)).sum().backward()
        g = G_ii.detach().cpu().numpy()
        want_g = z[f"{n}/g_ii"]
        _close(np.stack([g.real, g.imag]), np.stack([want_g.real, want_g.imag]), f"{n}: g_ii")
        _close(y.detach().float().cpu().numpy(), z[f"{n}/y"], f"{n}: y")
        params = dict(layer.named_parameters())
        for k in GRADS:
            assert params[k].grad is not None, (n, k)
            _close(params[k].grad.float().cpu().numpy().reshape(z[f"{n}/grad/{k}"].shape), z[f"{n}/grad/{k}"],
                   f"{n}: grad {k}")
        for k in ("output_proj.weight", "output_proj.bias", "bk_scale", "gamma", "epsilon_param"):
            assert params[k].grad is None or float(params[k].grad.abs().max()) == 0.0, (n, k)


def test_hybrid_layer_with_oracle_operator_matches_reference_layer():
    _run("cpu", oracle=True)


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, True])
def test_hybrid_layer_on_gpu_matches_reference_layer(fused):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    _run("cuda", fused=fused)


def _small_lm(hybrid_seed=5):
    from bkcore.lm import SmallResNetBKLM
    torch.manual_seed(hybrid_seed)
    m = SmallResNetBKLM(hybrid=True)                        # phase7_small dimensions (configs/phase7_small.yaml:7-12)
    with torch.no_grad():
        for i, blk in enumerate(m.blocks):
            blk.bk_layer.v_proj.weight.normal_(0.0, 1.0 / 16.0)          # a potential of order 1
            blk.bk_layer.h0_diag_base.fill_(-2.0)
            blk.bk_layer.h0_sub_base.fill_(1.0)
            blk.bk_layer.h0_super_base.fill_(1.0)
            blk.bk_layer.hybrid_attn.gate_sensitivity.fill_(1.0 + 0.5 * i)
    return m


@pytest.mark.gpu
def test_hybrid_lm_step_at_phase7_small_batch(tmp_path):
    """The whole small LM with the hybrid consumer at the configuration's own batch (8 x 256 tokens,
    configs/phase7_small.yaml:27): HIP operator on the GPU against the same model with the CPU oracle as the operator.
    Loss 1e-4 relative; the norm of every parameter's gradient 5e-3 relative (matrix products accumulate in a different
    order on the GPU; bfloat16 operands as in the reference)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    g = torch.Generator().manual_seed(77)
    tok = torch.randint(0, 50257, (8, 256), generator=g)
    tgt = torch.randint(0, 50257, (8, 256), generator=g)
    res = {}
    for dev in ("cpu", "cuda"):
        m = _small_lm().to(dev)
        if dev == "cpu":
            m.set_bk_core(_OracleOp.apply)
        logits = m(tok.to(dev))
        loss = torch.nn.functional.cross_entropy(logits.reshape(-1, logits.shape[-1]), tgt.to(dev).reshape(-1))
        loss.backward()
        res[dev] = (float(loss.detach()), {k: (None if p.grad is None else float(p.grad.double().norm()))
                                  for k, p in m.named_parameters()})
    assert abs(res["cuda"][0] - res["cpu"][0]) <= 1e-4 * res["cpu"][0]
    seen_v = 0
    for k, ref in res["cpu"][1].items():
        got = res["cuda"][1][k]
        if ref is None or ref == 0.0:
            assert got is None or got == 0.0, k
            continue
        assert abs(got - ref) <= 5e-3 * ref, (k, got, ref)
        seen_v += k.endswith("v_proj.weight")
    assert seen_v == 4                                       # every block's potential projection got its gradient
