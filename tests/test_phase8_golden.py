"""The caller that hands the operator learned, position-varying off-diagonals with b != c -- the Green-function step of
``BKCoreHyperbolicIntegration`` (src/models/phase8/bk_core_hyperbolic.py:367-407) -- against operands, outputs and
gradients of the reference module itself (tests/golden/bkphase8_golden.npz, oracle/gen_golden_phase8.py).

Pinned here: three bfloat16 projections averaged over the model dimension -> he_diag, h0_super, h0_sub -> operator ->
(G_ii, features), and backward from a cotangent on the features into he_diag_proj; the off-diagonal projections get no
gradient, as in the reference (bk_core.py:384-385).

CPU: the CPU oracle plugged in as the operator.  GPU: the HIP operator.  Tolerances as for the other layer fixtures:
the bulk tightly, every element to a bfloat16 step (a product accumulated in another order moves a bfloat16 operand by
one step, 2^-8 of its size)."""
import os

import numpy as np
import pytest
import torch

from test_layer_golden import _close
from test_lm_step import _OracleOp

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bkphase8_golden.npz")
PARAMS = ("he_diag_proj.weight", "he_diag_proj.bias", "h0_super_proj.weight", "h0_super_proj.bias",
          "h0_sub_proj.weight", "h0_sub_proj.bias")


class _OracleOperator:
    """Stands where ``bkcore.layers`` looks up the operator; same call as the module makes (a, b, c, z, use_theta_only)."""
    @staticmethod
    def apply(a, b, c, z, *rest):
        assert rest in ((), (False,))
        feats, G = _OracleOp.apply(a, b, c, z)
        return feats, G


def test_fixture_holds_what_the_test_needs():
    z = np.load(GOLD, allow_pickle=False)
    names = [str(n) for n in z["__names__"]]
    assert len(names) >= 3
    for n in names:
        B, N, D = z[f"{n}/x"].shape
        assert z[f"{n}/he_diag"].shape == (B, N) and z[f"{n}/h0_super"].shape == (B, N - 1) == z[f"{n}/h0_sub"].shape
        assert str(z[f"{n}/operand_dtype"]) == "torch.bfloat16"
        assert z[f"{n}/g_ii"].shape == (B, N) and z[f"{n}/features"].shape == (B, N, 2)
        assert np.array_equal(z[f"{n}/features"], np.stack([z[f"{n}/g_ii"].real, z[f"{n}/g_ii"].imag], -1))
        assert not bool(z[f"{n}/grad_is_none/he_diag_proj.weight"])
        assert all(bool(z[f"{n}/grad_is_none/{p}.{k}"]) for p in ("h0_super_proj", "h0_sub_proj") for k in ("weight", "bias"))
    # at least one case has off-diagonals that differ from each other and along the row
    n = names[-1]
    assert np.abs(z[f"{n}/h0_super"] - z[f"{n}/h0_sub"]).max() > 0.05 and z[f"{n}/h0_super"].std() > 0.05


def test_oracle_reproduces_reference_outputs_on_reference_operands():
    """The operator's part alone: the reference's own operands in, its G_ii out (float32 values, the reference's
    complex128 recursion rounded to complex64: agreement to 1e-5 of the largest element)."""
    from oracle import bk_oracle_torch as O
    z = np.load(GOLD, allow_pickle=False)
    for n in (str(n) for n in z["__names__"]):
        a, b, c = (torch.from_numpy(z[f"{n}/{k}"]) for k in ("he_diag", "h0_super", "h0_sub"))
        zz = complex(*z[f"{n}/z"])
        feats, G = O.forward(a, b, c, zz)
        want = z[f"{n}/g_ii"]
        assert np.abs(G.numpy() - want).max() <= 1e-5 * np.abs(want).max(), n
        np.testing.assert_array_equal(feats.numpy(), np.stack([G.numpy().real, G.numpy().imag], -1))


def _run(device, oracle=False):
    import bkcore
    import bkcore.layers as layers
    z = np.load(GOLD, allow_pickle=False)
    for n in (str(n) for n in z["__names__"]):
        x = torch.from_numpy(z[f"{n}/x"]).to(device)
        D = x.shape[-1]
        zr, zi = (float(v) for v in z[f"{n}/z"])
        m = bkcore.LearnedTridiagonalGreen(D, z_real=zr, z_imag=zi)
        m.load_state_dict({**{k: torch.from_numpy(z[f"{n}/param/{k}"]) for k in PARAMS}, "z": m.z}, strict=True)
        m = m.to(device)
        seen = {}
        inner = _OracleOperator if oracle else layers.BKCoreFunction

        class _Spy:
            @staticmethod
            def apply(a, b, c, zz, *rest):
                seen.update(a=a.detach(), b=b.detach(), c=c.detach(), rest=rest)
                return inner.apply(a, b, c, zz, *rest)
        saved, layers.BKCoreFunction = layers.BKCoreFunction, _Spy
        try:
            G, feats = m.compute_green_function(x)
        finally:
            layers.BKCoreFunction = saved
        (feats * torch.from_numpy(z[f"{n}/w"]).to(device)).sum().backward()

        assert seen["a"].dtype == torch.bfloat16 and seen["rest"] == (False,)            # as the reference calls it
        for key, name in (("a", "he_diag"), ("b", "h0_super"), ("c", "h0_sub")):
            _close(seen[key].float().cpu().numpy(), z[f"{n}/{name}"], f"{n}: operand {name}")
        g, want = G.detach().cpu().numpy(), z[f"{n}/g_ii"]
        assert g.dtype == np.complex64 and feats.dtype == torch.float32
        _close(np.stack([g.real, g.imag]), np.stack([want.real, want.imag]), f"{n}: g_ii")
        _close(feats.detach().cpu().numpy(), z[f"{n}/features"], f"{n}: features")
        for k in ("he_diag_proj.weight", "he_diag_proj.bias"):
            grad = dict(m.named_parameters())[k].grad
            assert grad is not None, (n, k)
            _close(grad.float().cpu().numpy(), z[f"{n}/grad/{k}"], f"{n}: grad {k}")
        for p in (m.h0_super_proj, m.h0_sub_proj):
            assert p.weight.grad is None and p.bias.grad is None, n


def test_learned_tridiagonal_green_with_oracle_operator_matches_reference_module():
    _run("cpu", oracle=True)


@pytest.mark.gpu
def test_learned_tridiagonal_green_on_gpu_matches_reference_module():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    _run("cuda")
