"""BKCoreLayer (plain and fused) against outputs and gradients of the reference's own MoEResNetBKLayer with its FFN
branch removed (tests/golden/bklayer_golden.npz, written by oracle/gen_golden_layer.py in the build container)."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bklayer_golden.npz")
KEEP = ("v_proj.weight", "v_proj.bias", "output_proj.weight", "output_proj.bias", "bk_scale", "gamma", "epsilon_param",
        "h0_diag_base", "h0_sub_base", "h0_super_base")


def _load():
    z = np.load(GOLD, allow_pickle=False)
    return z, [str(n) for n in z["__names__"]]


def test_layer_golden_file_is_complete():
    z, names = _load()
    assert len(names) >= 4
    for n in names:
        for k in ("x", "w", "y", "grad_x", "g_ii"):
            assert f"{n}/{k}" in z.files
        for k in KEEP:
            assert f"{n}/param/{k}" in z.files
        assert f"{n}/grad/v_proj.weight" in z.files and f"{n}/grad/bk_scale" in z.files


def _close(got, want, what):
    """The reference evaluates tanh on the CPU and the operator in complex128: a last-bit difference can move an
    element of the bfloat16 diagonal (or of the bfloat16-rounded gradient) by one step, i.e. 0.4 % of one element,
    so the bulk must agree tightly and every element loosely."""
    got, want = got.astype(np.float64), want.astype(np.float64)
    scale = np.abs(want).max() + 1e-30
    err = np.abs(got - want) / scale
    assert np.median(err) <= 2e-5, (what, float(np.median(err)))
    assert err.max() <= 2e-2, (what, float(err.max()))


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, True])
def test_layer_matches_reference_layer(fused):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import bkcore
    z, names = _load()
    for n in names:
        x0 = torch.from_numpy(z[f"{n}/x"])
        B, N, D = x0.shape
        layer = bkcore.BKCoreLayer(D, N, fused=fused)
        layer.load_state_dict({k: torch.from_numpy(z[f"{n}/param/{k}"]) for k in KEEP})
        layer = layer.cuda()
        x = x0.cuda().requires_grad_(True)
        y = layer(x)
        (y * torch.from_numpy(z[f"{n}/w"]).cuda()).sum().backward()
        _close(y.detach().cpu().numpy(), z[f"{n}/y"], f"{n}: y")
        g = layer.last_g_ii.detach().cpu().numpy()
        want_g = z[f"{n}/g_ii"]
        _close(np.stack([g.real, g.imag]), np.stack([want_g.real, want_g.imag]), f"{n}: g_ii")
        _close(x.grad.cpu().numpy(), z[f"{n}/grad_x"], f"{n}: grad_x")
        for k, p in layer.named_parameters():
            key = f"{n}/grad/{k}"
            if key in z.files:
                assert p.grad is not None, (n, k)
                _close(p.grad.cpu().numpy(), z[key], f"{n}: grad {k}")
            else:
                assert p.grad is None or float(p.grad.abs().max()) == 0.0, (n, k)
