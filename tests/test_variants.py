"""The reference's other backward wrappers (src/models/batched_gradient.py:12-217, src/kernels/bk_scan_stable.py:556-611)
as variants of the elementwise backward kernel.  Expected values were recorded from the imported reference by
oracle/gen_golden_variants.py (tests/golden/variants_golden.npz).  CPU: the oracle's restatements against the fixture;
GPU: the kernel and the autograd classes against it.  Tolerance: 2e-5 relative to max(|ref|, 1) per entry (float32
elementwise arithmetic; entries where the reference itself returns a non-finite number are compared as non-finite)."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "variants_golden.npz")


def _close(x, ref):
    x, ref = np.asarray(x, np.float64), np.asarray(ref, np.float64)
    fin = np.isfinite(ref)
    assert not np.isfinite(x[~fin]).any()
    assert np.all(np.abs(x[fin] - ref[fin]) <= 2e-5 * np.maximum(np.abs(ref[fin]), 1.0))


def test_oracle_restatements_match_the_reference():
    from oracle import bk_oracle_torch as O
    d = np.load(GOLD)
    G, gf, gG = torch.from_numpy(d["G"]), torch.from_numpy(d["gfeat"]), torch.from_numpy(d["gG"])
    for alpha in (0.0, 0.5, 1.0):
        _close(O.backward_batched(G, gf, alpha).numpy(), d[f"batched_a{alpha}"])
    _close(O.backward_batched(G, gf, 0.25).numpy(), d["batched_cls_a0.25"])
    _close(O.backward_batched(G, gf, 0.25).numpy(), d["memopt_cls_a0.25"])
    _close(O.backward_scan_stable(G, gG).numpy(), d["scan_stable_grad_a"])


@pytest.mark.gpu
def test_kernel_variants_match_the_reference():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import bkcore
    from bkcore import _lib
    d = np.load(GOLD)
    G, gf, gG = (torch.from_numpy(d[k]).cuda() for k in ("G", "gfeat", "gG"))
    for alpha in (0.0, 0.5, 1.0):
        _close(bkcore.bk_backward_variant(G, gf, None, alpha, _lib.BWD_BATCHED).cpu().numpy(), d[f"batched_a{alpha}"])
    _close(bkcore.bk_backward_variant(G, None, gG, 0.0, _lib.BWD_SCAN_STABLE).cpu().numpy(), d["scan_stable_grad_a"])
    # variant 0 is the operator's own reference-mode backward
    a = bkcore.bk_backward_variant(G, gf, gG, 0.5, _lib.BWD_REFERENCE)
    b = bkcore.bk_backward_reference(G, gf, gG, 0.5)
    assert torch.equal(torch.nan_to_num(a), torch.nan_to_num(b))
    L = _lib.load()
    out = torch.empty(4, device="cuda")
    assert L.bkcore_backward_variant(G.data_ptr(), None, None, 0.5, 4, out.data_ptr(), 7, None) == _lib.EINVAL


@pytest.mark.gpu
def test_autograd_classes_of_the_variants():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import bkcore
    from oracle import bk_oracle_torch as O
    g = torch.Generator().manual_seed(5)
    B, N, z = 3, 300, 1j
    a = (-2.0 + 0.5 * torch.randn(B, N, generator=g))
    b = 1.0 + 0.1 * torch.randn(B, N - 1, generator=g)
    c = 1.0 + 0.1 * torch.randn(B, N - 1, generator=g)
    gf = torch.randn(B, N, 2, generator=g)
    Gref = O.get_diag(a.double(), b.double(), c.double(), z).to(torch.complex64)     # no input clamps: get-diag semantics
    for cls, alpha in ((bkcore.BatchedAnalyticBKCoreFunction, 0.5), (bkcore.MemoryOptimizedBKCoreFunction, 0.3)):
        cls.GRAD_BLEND = alpha
        try:
            ad = a.cuda().requires_grad_()
            feats = cls.apply(ad, b.cuda(), c.cuda(), torch.tensor(z, dtype=torch.complex64, device="cuda"))
            assert feats.shape == (B, N, 2)
            np.testing.assert_allclose(torch.view_as_complex(feats.detach().contiguous()).cpu().numpy(), Gref.numpy(),
                                       rtol=1e-4, atol=1e-6)
            feats.backward(gf.cuda())
            ref = O.backward_batched(torch.view_as_complex(feats.detach().contiguous()).cpu(), gf, alpha)
            _close(ad.grad.cpu().numpy(), ref.numpy())
        finally:
            cls.GRAD_BLEND = 0.5
    ad = a.cuda().requires_grad_()
    bd = b.cuda().requires_grad_()
    cd = c.cuda().requires_grad_()
    G = bkcore.BKScanStableFunction.apply(ad, bd, cd, 0.0, 1.0)
    assert G.dtype == torch.complex64
    np.testing.assert_allclose(G.detach().cpu().numpy(), Gref.numpy(), rtol=1e-4, atol=1e-6)
    gG = torch.complex(gf[..., 0], gf[..., 1])
    G.backward(gG.cuda())
    _close(ad.grad.cpu().numpy(), O.backward_scan_stable(G.detach().cpu(), gG).numpy())
    assert float(bd.grad.abs().max()) == 0.0 and float(cd.grad.abs().max()) == 0.0
