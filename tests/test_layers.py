"""nn.Module mirrors of the reference's BK-Core call sites (construction on CPU; forward/backward on the GPU)."""
import numpy as np
import pytest
import torch


def test_layer_state_dict_names_match_reference():
    import bkcore
    layer = bkcore.BKCoreLayer(d_model=16, n_seq=32)
    keys = set(layer.state_dict().keys())
    # the BK-Core related entries of MoEResNetBKLayer.state_dict() (reference resnet_bk.py:84-87,124-127)
    assert keys == {"v_proj.weight", "v_proj.bias", "output_proj.weight", "output_proj.bias", "bk_scale", "gamma",
                    "epsilon_param", "h0_diag_base", "h0_sub_base", "h0_super_base"}
    assert layer.h0_diag_base.shape == (1, 32) and layer.h0_sub_base.shape == (1, 31)
    assert float(layer.v_proj.weight.std()) < 1e-3
    core = bkcore.CUDAOptimizedBKCore(n_seq=32)
    assert set(core.state_dict().keys()) == {"h0_diag", "h0_sub", "h0_super"}
    assert torch.all(core.h0_diag == -2.0) and torch.all(core.h0_super == 1.0) and core.z == 1j


def test_layer_refuses_cpu_tensors():
    import bkcore
    layer = bkcore.BKCoreLayer(d_model=8, n_seq=16)
    with pytest.raises(RuntimeError, match="ROCm device"):
        layer(torch.randn(2, 16, 8))


@pytest.mark.gpu
def test_layer_forward_backward_matches_oracle_composition():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import bkcore
    from oracle import bk_oracle_torch as O

    class RefOp(torch.autograd.Function):                        # oracle forward + reference-mode backward on the CPU
        @staticmethod
        def forward(ctx, a, b, c, z):
            feats, G = O.forward(a.float(), b.float(), c.float(), complex(z))
            ctx.save_for_backward(G)
            return feats, G

        @staticmethod
        def backward(ctx, gf, gG):
            (G,) = ctx.saved_tensors
            return O.backward_reference(G, gf, gG, 0.5), None, None, None

    torch.manual_seed(0)
    B, N, D = 3, 200, 16
    layer = bkcore.BKCoreLayer(D, N)
    with torch.no_grad():
        layer.v_proj.weight.normal_(0.0, 0.5)
        layer.h0_sub_base.fill_(0.4)
        layer.h0_super_base.fill_(0.6)
        layer.gamma.fill_(0.1)
    x = torch.randn(B, N, D)
    ref = bkcore.BKCoreLayer(D, N)
    ref.load_state_dict(layer.state_dict())
    ref.bk_core = RefOp.apply
    out_ref = ref(x)
    out_ref.sum().backward()
    dev = layer.cuda()
    out = dev(x.cuda())
    out.sum().backward()
    assert out.shape == (B, N, D)
    np.testing.assert_allclose(out.detach().cpu().numpy(), out_ref.detach().numpy(), rtol=2e-4, atol=2e-5)
    g, gr = dev.v_proj.weight.grad.cpu().numpy(), ref.v_proj.weight.grad.numpy()
    assert np.abs(g - gr).max() <= 2e-2 * np.abs(gr).max()       # grad_he_diag passes through bf16 (2^-8) on both sides
    assert dev.last_g_ii.shape == (B, N, 1) and dev.last_g_ii.dtype == torch.complex64


@pytest.mark.gpu
def test_cuda_optimized_bk_core_signature():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import bkcore
    torch.manual_seed(1)
    B, N = 4, 64
    core = bkcore.CUDAOptimizedBKCore(N).cuda()
    v = 0.3 * torch.randn(B, N)
    feats = core(v.cuda())
    assert feats.shape == (B, N, 2) and feats.dtype == torch.float32
    T = torch.zeros(B, N, N, dtype=torch.complex128)
    i = torch.arange(N)
    T[:, i, i] = (-2.0 + v.double()).to(torch.complex128) - 1j
    T[:, i[:-1], i[1:]] = 1.0
    T[:, i[1:], i[:-1]] = 1.0
    Gd = torch.diagonal(torch.linalg.inv(T), dim1=-2, dim2=-1)
    ref = torch.stack([Gd.real, Gd.imag], -1).float()
    # un-regularised resolvent vs the operator's regularised one: 1e-3 (the reference's own bound, tests/test_bk_core.py:47-77)
    np.testing.assert_allclose(feats.cpu().numpy(), ref.numpy(), atol=2e-3)


def test_learned_tridiagonal_green_names():
    import bkcore
    m = bkcore.LearnedTridiagonalGreen(d_model=8)
    assert set(m.state_dict().keys()) == {"he_diag_proj.weight", "he_diag_proj.bias", "h0_super_proj.weight",
                                          "h0_super_proj.bias", "h0_sub_proj.weight", "h0_sub_proj.bias", "z"}
    assert torch.all(m.he_diag_proj.bias == 1.0) and torch.all(m.h0_sub_proj.bias == 0.0)
    assert m.z == complex(0.0, 0.1)


@pytest.mark.gpu
def test_learned_offdiagonals_receive_gradients_in_analytic_mode():
    """The hyperbolic block's Green-function step (reference phase8/bk_core_hyperbolic.py:367-407) with position-varying
    learned b != c.  Reference mode: no gradient reaches the off-diagonal projections (bk_core.py:384-385).  Analytic
    mode: they get the gradient of the forward; checked against the float64 adjoint specification for d/d he_diag,
    d/d h0_super, d/d h0_sub on the operator's own (bf16-rounded) operands."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import bkcore
    from oracle.bk_analytic_adjoint import adjoint_grads
    torch.manual_seed(2)
    B, N, D = 2, 96, 8
    m = bkcore.LearnedTridiagonalGreen(D, z_imag=0.6).cuda()
    with torch.no_grad():
        m.he_diag_proj.weight.normal_(0.0, 0.2)
        m.h0_super_proj.weight.normal_(0.0, 0.2); m.h0_super_proj.bias.fill_(0.5)
        m.h0_sub_proj.weight.normal_(0.0, 0.2); m.h0_sub_proj.bias.fill_(0.7)
    x = torch.randn(B, N, D, device="cuda")
    gf = torch.randn(B, N, 2, device="cuda")

    G, feats = m(x)
    (feats * gf).sum().backward()
    assert m.he_diag_proj.weight.grad is not None
    assert m.h0_super_proj.weight.grad is None and m.h0_sub_proj.weight.grad is None     # reference behaviour

    m.zero_grad(set_to_none=True)
    bkcore.BKCoreFunction.GRAD_MODE = "analytic"
    try:
        captured = {}
        orig = bkcore.BKCoreFunction.apply

        def spy(a, b, c, z, *rest):                               # keep the operands to differentiate them directly
            a.retain_grad(); b.retain_grad(); c.retain_grad()
            captured.update(a=a, b=b, c=c)
            return orig(a, b, c, z, *rest)
        bkcore.layers.BKCoreFunction.apply, saved = spy, bkcore.layers.BKCoreFunction.apply
        try:
            G, feats = m(x)
            (feats * gf).sum().backward()
        finally:
            bkcore.layers.BKCoreFunction.apply = saved
    finally:
        bkcore.BKCoreFunction.GRAD_MODE = "reference"
    for proj in (m.he_diag_proj, m.h0_super_proj, m.h0_sub_proj):
        assert proj.weight.grad is not None and torch.isfinite(proj.weight.grad).all() and proj.weight.grad.abs().max() > 0
    a, b, c = (captured[k].detach().double().cpu().numpy() for k in "abc")
    gbar = torch.complex(gf[..., 0], gf[..., 1]).to(torch.complex128).cpu().numpy()
    ra, rb, rc, _ = adjoint_grads(a, b, c, complex(0.0, 0.6), gbar, "reference")
    for got, ref in ((captured["a"].grad, ra), (captured["b"].grad, rb), (captured["c"].grad, rc)):
        got = got.double().cpu().numpy()
        assert np.abs(got - ref).max() <= 1e-2 * np.abs(ref).max()   # the operand gradients are rounded to bf16 (2^-8)
