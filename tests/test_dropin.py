"""The dispatching drop-in (bkcore.dropin): CPU tensors reach the reference's own definitions, GPU tensors the HIP
operator, and the reference's class attributes are forwarded to both.  In these tests the CPU oracle stands in for the
reference's PyTorch body (the reference cannot travel with the tests); inputs are the reference's own N = 4
known-answer case (tests/test_bk_core.py:21-45: a = -2, b = c = 1, z = 1j), expected values from the golden fixture
recorded from the imported reference."""
import warnings

import numpy as np
import pytest
import torch


class _RefStandIn(torch.autograd.Function):
    """Same interface as the reference's BKCoreFunction (bk_core.py:219-246): apply(he_diag, h0_super, h0_sub, z,
    use_triton=None, use_parallel_scan=None), class attributes GRAD_BLEND / USE_TRITON / USE_PARALLEL_SCAN."""
    GRAD_BLEND = 0.5
    USE_TRITON = None
    USE_PARALLEL_SCAN = True
    calls = 0

    @staticmethod
    def forward(ctx, a, b, c, z, use_triton=None, use_parallel_scan=None):
        from oracle import bk_oracle_torch as O
        _RefStandIn.calls += 1
        feats, G = O.forward(a.float(), b.float(), c.float(), complex(z))
        ctx.save_for_backward(G)
        return feats, G

    @staticmethod
    def backward(ctx, gf, gG):
        from oracle import bk_oracle_torch as O
        (G,) = ctx.saved_tensors
        return O.backward_reference(G, gf, gG, _RefStandIn.GRAD_BLEND), None, None, None, None, None


def _ref_vmapped(a, b, c, z):
    from oracle import bk_oracle_torch as O
    return O.forward(a.float(), b.float(), c.float(), complex(z))[1]


def _ref_single(a, b, c, z):
    return _ref_vmapped(a[None], b[None], c[None], z)[0]


def _shim():
    from bkcore.dropin import make_dispatching
    return make_dispatching(_RefStandIn, _ref_vmapped, _ref_single)


def test_cpu_tensors_reach_the_reference_definitions(golden):
    F, vmapped, single = _shim()
    cs = golden["kat4"]
    a = torch.tensor([-2.0, -2.0, -2.0, -2.0])
    b = torch.tensor([1.0, 1.0, 1.0])
    c = torch.tensor([1.0, 1.0, 1.0])
    z = torch.tensor(1.0j, dtype=torch.complex64)
    before = _RefStandIn.calls
    G = single(a, b, c, z)
    assert G.shape == (4,) and G.dtype == torch.complex64
    np.testing.assert_allclose(G.numpy(), cs["G"][0], rtol=1e-5, atol=1e-7)
    ad = a[None].clone().requires_grad_()
    feats, G2 = F.apply(ad, b[None], c[None], z)
    assert _RefStandIn.calls == before + 1                         # the autograd class of the reference ran, on the CPU
    np.testing.assert_allclose(feats.detach().numpy(), cs["features"], rtol=1e-5, atol=1e-7)
    gf = torch.from_numpy(cs["gfeat"]).float()
    (feats * gf).sum().backward()
    np.testing.assert_allclose(ad.grad.numpy(), cs["grad_ref_a0.5_gG0"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(vmapped(a[None], b[None], c[None], z).numpy(), cs["G"], rtol=1e-5, atol=1e-7)
    feats3, _ = F.apply(a[None], b[None], c[None], z, False, False)  # the optional flags of the reference's signature
    assert torch.equal(feats3, feats.detach())


def test_class_attributes_are_forwarded_to_both_implementations():
    import bkcore
    F, _, _ = _shim()
    old = bkcore.BKCoreFunction.GRAD_BLEND
    try:
        assert F.GRAD_BLEND == 0.5                                   # read: the reference class
        F.GRAD_BLEND = 0.25                                          # write: both (scripts set it, bk_core.py:366 reads it)
        assert _RefStandIn.GRAD_BLEND == 0.25 and bkcore.BKCoreFunction.GRAD_BLEND == 0.25 and F.GRAD_BLEND == 0.25
        F.USE_PARALLEL_SCAN = False
        assert _RefStandIn.USE_PARALLEL_SCAN is False and bkcore.BKCoreFunction.USE_PARALLEL_SCAN is False
    finally:
        _RefStandIn.GRAD_BLEND = 0.5
        _RefStandIn.USE_PARALLEL_SCAN = True
        bkcore.BKCoreFunction.GRAD_BLEND = old
        bkcore.BKCoreFunction.USE_PARALLEL_SCAN = True


@pytest.mark.gpu
def test_gpu_tensors_reach_the_hip_operator(golden):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import bkcore
    F, vmapped, single = _shim()
    cs = golden["varying_bc"]
    a, b, c = (torch.from_numpy(np.ascontiguousarray(cs[k])).float().cuda() for k in ("a", "b", "c"))
    z = complex(*cs["z"])
    before = _RefStandIn.calls
    feats, G = F.apply(a, b, c, z)
    assert _RefStandIn.calls == before and G.is_cuda
    f2, G2 = bkcore.BKCoreFunction.apply(a, b, c, z)
    assert torch.equal(G, G2) and torch.equal(feats, f2)
    assert torch.equal(vmapped(a, b, c, z), bkcore.vmapped_get_diag(a, b, c, z))
    assert torch.equal(single(a[0], b[0], c[0], z), bkcore.get_tridiagonal_inverse_diagonal(a[0], b[0], c[0], z))
    F.apply(a.cpu(), b.cpu(), c.cpu(), z)                            # and CPU tensors still go to the reference side
    assert _RefStandIn.calls == before + 1


@pytest.mark.gpu
def test_lazy_nonfinite_gradient_warning():
    """WARN_NONFINITE_GRAD = "lazy": the reference's warning (bk_core.py:321-325) without a synchronisation inside
    backward; it is emitted once the asynchronous read-back has landed."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import bkcore
    g = torch.Generator(device="cuda").manual_seed(3)
    a = (-2.0 + 0.5 * torch.randn(4, 1001, device="cuda", generator=g)).requires_grad_()
    b = torch.ones(4, 1000, device="cuda")
    bkcore.BKCoreFunction.WARN_NONFINITE_GRAD = "lazy"
    try:
        feats, _ = bkcore.BKCoreFunction.apply(a, b, b, 1j)
        gf = torch.randn(4, 1001, 2, device="cuda", generator=g)
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            feats.backward(gf, retain_graph=True)
            assert not bkcore.poll_nonfinite_grad_warning(wait=True)
            assert not [w for w in rec if "NaN or Inf" in str(w.message)]
        clean = a.grad.clone()
        a.grad = None
        gf_bad = gf.clone()
        gf_bad[2, 1000, 1] = float("nan")                             # the last element: the kernel's tail path
        gf_bad[1, 17, 0] = float("inf")
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            feats.backward(gf_bad, retain_graph=True)
            assert bkcore.poll_nonfinite_grad_warning(wait=True)
            assert [w for w in rec if "NaN or Inf" in str(w.message)]
        # the values: NaN counts as 0, +inf as the largest float (torch.nan_to_num, bk_core.py:325)
        ref = torch.nan_to_num(gf_bad)
        a.grad = None
        feats.backward(ref)
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            assert not bkcore.poll_nonfinite_grad_warning(wait=True)   # the flag was cleared after it was reported
        assert torch.isfinite(clean).all()
    finally:
        bkcore.BKCoreFunction.WARN_NONFINITE_GRAD = False
