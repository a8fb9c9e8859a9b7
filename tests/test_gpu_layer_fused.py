"""The layer's elementwise steps fused around the operator (bkcore_layer_prologue, bkcore_layer_backward_ref through
the C ABI) against the same chain written with tensor-library ops (reference src/models/resnet_bk.py:139-170,216-217)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def _unfused_he_diag(v_pre, h0_diag, v_max):
    v = torch.nan_to_num(v_pre, nan=0.0, posinf=v_max, neginf=-v_max)
    v = v_max * torch.tanh(v / v_max)
    return (h0_diag.reshape(1, -1).expand(v.shape[0], -1) + v).to(torch.bfloat16)


@pytest.mark.parametrize("B,N", [(1, 1), (3, 17), (8, 256), (5, 4097), (64, 1024), (2, 5296), (3, 20000), (2, 32768)])
def test_prologue_matches_tensor_ops(B, N):
    _need_gpu()
    import bkcore
    g = torch.Generator(device="cuda").manual_seed(B * 1000 + N)
    v = 4.0 * torch.randn(B, N, generator=g, device="cuda")
    h0 = torch.randn(N, generator=g, device="cuda")
    if N >= 17:
        v[0, 3] = float("nan")
        v[-1, 5] = float("inf")
        v[0, 7] = float("-inf")
        v[-1, 9] = 1e30
    want = _unfused_he_diag(v, h0, 5.0)
    got = bkcore.layer_prologue(v, h0, 5.0, torch.bfloat16)
    assert got.dtype == torch.bfloat16 and got.shape == (B, N)
    # same arithmetic; a last-bit difference of the library tanh may move a value across a bfloat16 rounding
    # boundary: at most one bfloat16 step on at most 0.1 % of the elements
    d = (got.float() - want.float()).abs()
    step = want.float().abs().clamp_min(1e-30) * 2.0 ** -7
    assert bool((d <= step).all())
    assert float((d > 0).float().mean()) <= 1e-3
    got32 = bkcore.layer_prologue(v, h0, 5.0, torch.float32)
    vv = torch.nan_to_num(v, nan=0.0, posinf=5.0, neginf=-5.0)
    want32 = h0.reshape(1, -1) + 5.0 * torch.tanh(vv / 5.0)
    torch.testing.assert_close(got32, want32, rtol=1e-6, atol=1e-6)


def test_prologue_strided_rows_and_errors():
    _need_gpu()
    import bkcore
    big = torch.randn(6, 300, device="cuda")
    v = big[:, :200]                                   # row stride 300
    h0 = torch.full((1, 200), -2.0, device="cuda")
    got = bkcore.layer_prologue(v, h0, 5.0)
    want = _unfused_he_diag(v.contiguous(), h0, 5.0)
    assert float(((got.float() - want.float()).abs() > 0).float().mean()) <= 1e-3
    with pytest.raises(ValueError):
        bkcore.layer_prologue(v, torch.zeros(199, device="cuda"), 5.0)
    with pytest.raises(RuntimeError, match="invalid argument"):
        bkcore.layer_prologue(v, h0, 0.0)
    with pytest.raises(RuntimeError, match="ROCm device"):
        bkcore.layer_prologue(torch.zeros(2, 4), torch.zeros(4), 5.0)


def _chain_unfused(v_pre, h0_diag, b, c, z, v_max, fclamp):
    import bkcore
    B = v_pre.shape[0]
    v = torch.nan_to_num(v_pre, nan=0.0, posinf=v_max, neginf=-v_max)
    v = v_max * torch.tanh(v / v_max)
    he = h0_diag.expand(B, -1) + v
    feats, G = bkcore.BKCoreFunction.apply(he.to(torch.bfloat16), b.expand(B, -1).to(torch.bfloat16),
                                           c.expand(B, -1).to(torch.bfloat16), z)
    if fclamp is not None:
        feats = torch.clamp(feats, -fclamp, fclamp)
    return feats, G


@pytest.mark.parametrize("B,N,fclamp,with_gG", [(4, 64, 5.0, False), (8, 256, 0.3, True), (2, 4500, None, False),
                                                (3, 33, 0.3, True), (3, 1, 5.0, False)])
def test_fused_chain_forward_backward(B, N, fclamp, with_gG):
    _need_gpu()
    import bkcore
    g = torch.Generator(device="cuda").manual_seed(N)
    v0 = 2.0 * torch.randn(B, max(N, 3), generator=g, device="cuda")
    v0[0, 1] = float("nan")
    v0[-1, 2] = float("inf")
    v0 = v0[:, :N].contiguous()
    h0 = torch.full((1, N), -2.0, device="cuda") + 0.1 * torch.randn(1, N, generator=g, device="cuda")
    b = 1.0 + 0.1 * torch.randn(1, N - 1, generator=g, device="cuda")
    c = 1.0 + 0.1 * torch.randn(1, N - 1, generator=g, device="cuda")
    z = torch.tensor(0.2 + 0.9j, dtype=torch.complex64, device="cuda")
    gf = torch.randn(B, N, 2, generator=g, device="cuda")
    gG = torch.complex(torch.randn(B, N, generator=g, device="cuda"), torch.randn(B, N, generator=g, device="cuda"))

    va = v0.clone().requires_grad_(True)
    fa, Ga = _chain_unfused(va, h0, b, c, z, 5.0, fclamp)
    vb = v0.clone().requires_grad_(True)
    fb, Gb = bkcore.FusedBKLayerFunction.apply(vb, h0, b, c, z, 5.0, fclamp)
    # rows are independent systems: where a row's bfloat16 diagonal is identical the same operator kernel ran on the
    # same numbers; a last-bit difference of tanh can move an element across a bfloat16 rounding boundary
    he_a = _unfused_he_diag(v0, h0, 5.0)
    he_b = bkcore.layer_prologue(v0, h0, 5.0)
    same = (he_a == he_b).all(dim=1)
    assert float((he_a != he_b).float().mean()) <= 1e-3
    assert torch.equal(Ga[same], Gb[same]) and torch.equal(fa[same], fb[same])
    torch.testing.assert_close(torch.view_as_real(Gb), torch.view_as_real(Ga), rtol=2e-2, atol=2e-2)
    if fclamp is not None:
        assert float(fb.detach().abs().max()) <= float(torch.tensor(fclamp, dtype=torch.float32))
    outs_a, outs_b = ((fa, Ga), (fb, Gb)) if with_gG else ((fa,), (fb,))
    cot = (gf, gG) if with_gG else (gf,)
    torch.autograd.backward(outs_a, cot)
    torch.autograd.backward(outs_b, cot)
    ga, gb = va.grad, vb.grad
    assert gb.dtype == torch.float32 and gb.shape == (B, N)
    if N >= 3:
        assert float(gb[0, 1]) == 0.0 and float(gb[-1, 2]) == 0.0      # non-finite v_pre: no gradient
    # same formula; a last-bit difference before the bfloat16 rounding of the operator's gradient can move one step
    d = (ga - gb).abs()
    tol = 2.0 ** -7 * ga.abs() + 1e-6
    assert bool((d <= tol)[same].all()), float((d - tol)[same].max())
    assert float((d > 1e-6 * ga.abs() + 1e-9)[same].float().mean()) <= 5e-3
    torch.testing.assert_close(gb, ga, rtol=5e-2, atol=5e-2 * float(ga.abs().max()))

def test_layer_fused_option_matches_unfused_layer():
    _need_gpu()
    import bkcore
    torch.manual_seed(0)
    D, N, B = 32, 128, 4
    la = bkcore.BKCoreLayer(D, N).cuda()
    lb = bkcore.BKCoreLayer(D, N, fused=True).cuda()
    lb.load_state_dict(la.state_dict())
    with torch.no_grad():
        for l in (la, lb):
            l.v_proj.weight.mul_(3000.0)                # make the potential matter
    lb.load_state_dict(la.state_dict())
    x = torch.randn(B, N, D, device="cuda")
    xa = x.clone().requires_grad_(True)
    xb = x.clone().requires_grad_(True)
    ya, yb = la(xa), lb(xb)
    torch.testing.assert_close(yb, ya, rtol=2e-2, atol=2e-3)
    w = torch.randn_like(ya)
    (ya * w).sum().backward()
    (yb * w).sum().backward()
    torch.testing.assert_close(xb.grad, xa.grad, rtol=2e-2, atol=1e-4 * float(xa.grad.abs().max()) + 1e-9)
    for (n1, p1), (_, p2) in zip(la.named_parameters(), lb.named_parameters()):
        if p1.grad is None:
            assert p2.grad is None, n1
            continue
        torch.testing.assert_close(p2.grad, p1.grad, rtol=2e-2, atol=1e-3 * float(p1.grad.abs().max()) + 1e-9, msg=n1)
    assert lb.last_g_ii.shape == (B, N, 1) and lb.last_g_ii.dtype == torch.complex64


@pytest.mark.parametrize("B,N,kind", [(6, 300, "good"), (3, 9000, "good"), (4, 40000, "good"), (4, 40000, "scan"),
                                      (5, 5000, "illcond"), (2, 1, "good")])
def test_forward_clamped_features(B, N, kind):
    """bkcore_forward_clamped: features = clamp(features of bkcore_forward), G untouched, on every forward path (one
    workgroup per row, several per row, scan first launch, rows recomputed by the second launch)."""
    _need_gpu()
    import bkcore
    from bkcore import bk_core
    g = torch.Generator(device="cuda").manual_seed(N + B)
    if kind == "illcond":
        a = torch.randn(B, N, generator=g, device="cuda")
        b = torch.randn(B, max(N - 1, 0), generator=g, device="cuda")
        c = torch.randn(B, max(N - 1, 0), generator=g, device="cuda")
        z = 0.1 + 0.1j
    else:
        a = -2.0 + 0.5 * torch.randn(B, N, generator=g, device="cuda")
        b = torch.ones(B, max(N - 1, 0), device="cuda")
        c = b
        z = 1j
    bk_core._DEBUG["scan_path"] = kind == "scan"
    try:
        f0, G0 = bkcore.bk_forward(a, b, c, z)
        f1, G1 = bkcore.bk_forward(a, b, c, z, feature_clamp=0.25)
    finally:
        bk_core._DEBUG["scan_path"] = False
    assert torch.equal(G0, G1)
    assert torch.equal(f1, torch.clamp(f0, -0.25, 0.25))
    assert float(f0.abs().max()) > 0.25                      # the clamp was active somewhere
    with pytest.raises(ValueError):
        bkcore.bk_forward(a, b, c, z, feature_clamp=0.0)


@pytest.mark.parametrize("lead,D,with_res", [((3, 50), 256, True), ((1, 1), 4, False), ((7, 33), 30, True),
                                             ((2, 200), 1024, False), ((1500,), 64, True)])
def test_spectral_output_matches_tensor_ops(lead, D, with_res):
    """bkcore_layer_epilogue / _backward against residual + bk_scale * Linear(2 -> D)(features) under autograd."""
    _need_gpu()
    import bkcore
    g = torch.Generator(device="cuda").manual_seed(D)
    mk = lambda *s: torch.randn(*s, generator=g, device="cuda")
    f0, W0, b0, s0 = mk(*lead, 2), mk(D, 2), mk(D), 1.0 + 0.1 * mk(D)
    r0 = mk(*lead, D) if with_res else None
    go = mk(*lead, D)

    def run(fn):
        leaves = [t.clone().requires_grad_(True) for t in (f0, W0, b0, s0)]
        res = r0.clone().requires_grad_(True) if with_res else None
        out = fn(*leaves, res)
        out.backward(go)
        return out.detach(), [t.grad for t in leaves] + ([res.grad] if with_res else [])

    def plain(f, W, b, s, res):
        y = s * torch.nn.functional.linear(f, W, b)
        return y if res is None else res + y

    out_a, grads_a = run(plain)
    out_b, grads_b = run(bkcore.FusedSpectralOutputFunction.apply)
    torch.testing.assert_close(out_b, out_a, rtol=1e-5, atol=1e-5)
    n = f0.numel() // 2
    for ga, gb, name in zip(grads_a, grads_b, ("features", "weight", "bias", "bk_scale", "residual")):
        assert gb.shape == ga.shape, name
        # sums of n (columns) or D (features) float32 products in a different order
        atol = 2e-6 * float(ga.abs().max()) * max(n, D) ** 0.5 + 1e-6
        torch.testing.assert_close(gb, ga, rtol=1e-4, atol=atol, msg=lambda m, name=name: f"{name}: {m}")
    # deterministic: the same call twice gives the same bits
    _, grads_c = run(bkcore.FusedSpectralOutputFunction.apply)
    assert all(torch.equal(x, y) for x, y in zip(grads_b, grads_c))


def test_spectral_output_limits():
    _need_gpu()
    import bkcore
    f = torch.randn(4, 2, device="cuda", requires_grad=True)
    W = torch.randn(2048, 2, device="cuda")
    with pytest.raises(ValueError, match="1024"):
        bkcore.FusedSpectralOutputFunction.apply(f, W, torch.zeros(2048, device="cuda"),
                                                 torch.ones(2048, device="cuda"), None)
    with pytest.raises(ValueError):
        bkcore.FusedSpectralOutputFunction.apply(f, W[:8, :1], torch.zeros(8, device="cuda"),
                                                 torch.ones(8, device="cuda"), None)
    with pytest.raises(RuntimeError, match="ROCm device"):
        bkcore.FusedSpectralOutputFunction.apply(torch.zeros(4, 2), torch.zeros(8, 2), torch.zeros(8), torch.ones(8), None)


@pytest.mark.gpu
@pytest.mark.parametrize("D", [256, 96])
def test_output_stage_reads_features_through_the_clamp(D):
    """FusedSpectralOutputFunction(raw, ..., feature_clamp=c) == the same function on torch.clamp(raw, -c, c)
    (resnet_bk.py:216-219), bit for bit, and the gradient for `raw` is that of the clamp composed with it: zero where
    |raw| > c (both the wave-per-position kernel, D = 256, and the general one)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import bkcore
    g = torch.Generator().manual_seed(D)
    n, c = 1500, 0.7
    raw = torch.randn(n, 2, generator=g).cuda()
    raw[3, 0] = c                      # on the boundary: the clamp's gradient is 1 there
    raw[5, 1] = -c
    lin = torch.nn.Linear(2, D).cuda()
    sc = (1.0 + 0.1 * torch.randn(D, generator=g)).cuda().requires_grad_(True)
    res = torch.randn(n, D, generator=g).cuda()
    go = torch.randn(n, D, generator=g).cuda()

    def run(through):
        x = raw.detach().clone().requires_grad_(True)
        lin.zero_grad(set_to_none=True); sc.grad = None
        if through:
            out = bkcore.FusedSpectralOutputFunction.apply(x, lin.weight, lin.bias, sc, res, c)
        else:
            out = bkcore.FusedSpectralOutputFunction.apply(torch.clamp(x, -c, c), lin.weight, lin.bias, sc, res)
        out.backward(go)
        return out.detach(), x.grad, lin.weight.grad.clone(), lin.bias.grad.clone(), sc.grad.clone()

    a, b = run(True), run(False)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert float(a[1][raw.abs() > c].abs().max()) == 0.0 and float(a[1][3, 0].abs()) > 0.0
