"""The HIP operator against the REFERENCE ITSELF at the headline sequence length (N = 32768, fixtures of
oracle/gen_golden_long.py: src/models/bk_core.py:27-216 forward, :303-385 backward, torch autograd through
vmapped_get_diag for the analytic mode), and the cotangent on G_ii alone -- what the hybrid-attention layer of
configs/phase7_small.yaml hands the operator (src/models/resnet_bk.py:172-185: the features are unused there)."""
import os

import numpy as np
import pytest
import torch

from conftest import ROOT, max_rel_err
from oracle.gen_golden_long import checksum, long_inputs
from test_oracle_golden import ALL

pytestmark = pytest.mark.gpu

TOL = 1e-4
CASES = ["lap", "varying"]


@pytest.fixture(scope="module")
def bk():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import bkcore
    return bkcore


@pytest.fixture(scope="module")
def golden_long():
    return np.load(os.path.join(ROOT, "tests", "golden", "bkcore_golden_long.npz"), allow_pickle=False)


def _inputs(golden_long, name):
    a, b, c, z, gf, gG = long_inputs(name)
    assert np.array_equal(checksum(a, b, c, gf, gG), golden_long[f"{name}/checksum"]), "seeded inputs differ"
    return a, b, c, z, gf, gG


@pytest.mark.parametrize("name", CASES)
def test_forward_matches_reference_at_headline_length(bk, golden_long, name):
    a, b, c, z, _, _ = _inputs(golden_long, name)
    ref = golden_long[f"{name}/G_raw"]
    zt = torch.tensor(z, dtype=torch.complex64, device="cuda")
    feats, G = bk.BKCoreFunction.apply(a.cuda(), b.cuda(), c.cuda(), zt)
    err = max_rel_err(G.cpu().numpy(), ref)
    print(f"{name}: max rel err vs the reference at N = 32768: {err:.2e}")
    assert err <= TOL
    assert torch.equal(feats, torch.view_as_real(G))
    Gr = bk.vmapped_get_diag(a.cuda(), b.cuda(), c.cuda(), z)
    assert max_rel_err(Gr.cpu().numpy(), ref) <= TOL
    # the same two rows inside a batch of the headline's row count (rows in two units of one launch)
    big = 1024
    A = a[:1].repeat(big, 1).cuda(); Bm = b[:1].repeat(big, 1).cuda(); Cm = c[:1].repeat(big, 1).cuda()
    A[5], Bm[5], Cm[5] = a[1].cuda(), b[1].cuda(), c[1].cuda()
    _, Gb = bk.BKCoreFunction.apply(A, Bm, Cm, zt)
    assert max_rel_err(Gb[[0, 5]].cpu().numpy(), ref) <= TOL


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("tag", ["gG0", "gG1", "gGonly"])
def test_reference_backward_at_headline_length(bk, golden_long, name, tag):
    """End to end through autograd: forward on the GPU, reference-mode backward with the cotangent of the fixture."""
    a, b, c, z, gf, gG = _inputs(golden_long, name)
    ref = golden_long[f"{name}/grad_ref_a0.5_{tag}"]
    ad = a.cuda().requires_grad_(True)
    bk.BKCoreFunction.GRAD_BLEND = 0.5
    feats, G = bk.BKCoreFunction.apply(ad, b.cuda(), c.cuda(), torch.tensor(z, dtype=torch.complex64, device="cuda"))
    loss = 0.0
    if tag != "gGonly":
        loss = loss + (feats * gf.cuda()).sum()
    if tag != "gG0":
        loss = loss + (G.real * gG.real.cuda() + G.imag * gG.imag.cuda()).sum()
    loss.backward()
    got = ad.grad.cpu().numpy()
    # The formula is -Re(conj(g) G^2) / 2 - Re(conj(g) / G^2) / 2 (bk_core.py:343-367): a relative error e in G moves
    # each term by 2 e of the TERM's size, whatever is left of the terms after their real parts cancel.  G is within
    # 2.8e-5 of the reference at this length (printed by the forward test), so the gradient is held to 1e-4 of the
    # size of its terms, elementwise; the kernel's own arithmetic is held to 1e-4 of the result just below.
    Gr = golden_long[f"{name}/G_raw"]
    gbar = np.zeros_like(Gr)
    if tag != "gGonly":
        gbar = gbar + (gf[..., 0] + 1j * gf[..., 1]).numpy()
    if tag != "gG0":
        gbar = gbar + gG.numpy()
    m2 = np.abs(Gr) ** 2
    terms = 0.5 * np.abs(gbar) * (m2 + 1.0 / np.maximum(m2, 1e-4))
    err = float(np.max(np.abs(got - ref) / (terms + 1e-5)))
    print(f"{name} {tag}: gradient error relative to its terms {err:.2e}")
    assert err <= TOL
    # and the backward kernel alone on the reference's own G: the elementwise formula
    f = None if tag == "gGonly" else gf.cuda()
    g = None if tag == "gG0" else gG.cuda()
    alone = bk.bk_backward_reference(torch.from_numpy(golden_long[f"{name}/G_raw"]).cuda(), f, g, 0.5)
    np.testing.assert_allclose(alone.cpu().numpy(), ref, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("name", CASES)
def test_analytic_backward_matches_reference_autograd(bk, golden_long, name):
    a, b, c, z, gf, _ = _inputs(golden_long, name)
    ga, gb, gc = bk.bk_backward_analytic(a.cuda(), b.cuda(), c.cuda(), z, gf.cuda(), None)
    for got, key in ((ga, "a"), (gb, "b"), (gc, "c")):
        ref = golden_long[f"{name}/grad_true_{key}"]
        assert np.abs(got.cpu().numpy() - ref).max() <= TOL * np.abs(ref).max()


@pytest.mark.parametrize("name", ALL)
@pytest.mark.parametrize("alpha", [0.0, 0.5, 1.0])
def test_golden_backward_G_only(bk, golden, name, alpha):
    """bk_backward_ref_kernel with a cotangent on G_ii and none on the features (bk_core.py:329-339), every fixture."""
    cs = golden[name]
    G = torch.from_numpy(np.ascontiguousarray(cs["G"])).cuda()
    gG = torch.from_numpy(np.ascontiguousarray(cs["gG"])).cuda()
    got = bk.bk_backward_reference(G, None, gG, alpha)
    np.testing.assert_allclose(got.cpu().numpy(), cs[f"grad_ref_a{alpha}_gGonly"], rtol=1e-4, atol=1e-5)


def test_autograd_with_only_G_used(bk, golden):
    """The autograd path where the loss touches G_ii alone: backward receives grad_features = None."""
    cs = golden["lap_n64_b8"]
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    a = dev(cs["a"]).requires_grad_(True)
    gG = dev(cs["gG"])
    for alpha in (0.0, 0.5, 1.0):
        bk.BKCoreFunction.GRAD_BLEND = alpha
        a.grad = None
        _, G = bk.BKCoreFunction.apply(a, dev(cs["b"]), dev(cs["c"]), torch.tensor(1j, dtype=torch.complex64, device="cuda"))
        (G.real * gG.real + G.imag * gG.imag).sum().backward()
        ref = cs[f"grad_ref_a{alpha}_gGonly"]
        got = a.grad.cpu().numpy()
        m2 = np.abs(cs["G"]) ** 2        # relative to the size of the formula's terms, as above (measured 4e-5)
        terms = np.abs(cs["gG"]) * ((1.0 - alpha) * m2 + alpha / np.maximum(m2, 1e-4))
        assert np.max(np.abs(got - ref) / (terms + 1e-5)) <= TOL
    bk.BKCoreFunction.GRAD_BLEND = 0.5
