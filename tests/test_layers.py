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
