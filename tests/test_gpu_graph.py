"""A forward captured in a HIP graph replays correctly: the verdicts that travel from a call's first launch to its
last one live in the device-side state block and clear themselves, so the NaN corner of the reference
(src/models/bk_core.py:260-264: NaNs in b, c are replaced only when he_diag holds a NaN) is decided per replay."""
import numpy as np
import pytest
import torch

from conftest import max_rel_err

pytestmark = pytest.mark.gpu


def _cases(B, N):
    g = torch.Generator().manual_seed(5)
    a = -2.0 + 0.5 * torch.randn(B, N, generator=g)
    b = torch.ones(B, N - 1)
    c = torch.ones(B, N - 1)
    clean = (a, b, c)
    a1, b1, c1 = a.clone(), b.clone(), c.clone()
    a1[0, 3] = float("nan"); a1[1, 7] = float("inf"); b1[0, 5] = float("nan"); c1[2, 2] = float("nan")
    b2 = b.clone(); b2[1, 40] = float("nan")
    # an ill-conditioned row: fails the first launch's checks and is recomputed by the last launch
    a3 = a.clone(); a3[3] = torch.randn(N, generator=g); b3 = b.clone(); b3[3] = torch.randn(N - 1, generator=g)
    return {"clean": clean, "poisoned": (a1, b1, c1), "nan_offdiag_only": (a, b2, c), "ill_row": (a3, b3, c)}


def test_graph_replay_matches_eager_and_oracle():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import bkcore
    from oracle import bk_oracle_c as C
    B, N = 4, 5000
    cases = _cases(B, N)
    z = torch.tensor(0.2 + 1j, dtype=torch.complex64, device="cuda")
    sa, sb, sc = (torch.empty(B, N, device="cuda"), torch.empty(B, N - 1, device="cuda"),
                  torch.empty(B, N - 1, device="cuda"))
    eager = {}
    for name, (a, b, c) in cases.items():
        eager[name] = bkcore.bk_forward(a.cuda(), b.cuda(), c.cuda(), z)[1].clone()
    sa.copy_(cases["clean"][0]); sb.copy_(cases["clean"][1]); sc.copy_(cases["clean"][2])
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            bkcore.bk_forward(sa, sb, sc, z)
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        feats, G = bkcore.bk_forward(sa, sb, sc, z)
    for name in ("clean", "poisoned", "nan_offdiag_only", "clean", "ill_row", "nan_offdiag_only", "poisoned", "clean"):
        a, b, c = cases[name]
        sa.copy_(a); sb.copy_(b); sc.copy_(c)
        graph.replay()
        torch.cuda.synchronize()
        got = G.cpu().numpy()
        # NaN-aware bitwise comparison with the eager call on the same input (same kernels)
        assert np.array_equal(got.view(np.uint32), eager[name].cpu().numpy().view(np.uint32)), name
        ref = C.forward_f32(a.numpy(), b.numpy(), c.numpy(), complex(0.2, 1.0))[1]
        if name == "ill_row":
            got, ref = got[:3], ref[:3]            # the ill-conditioned row has no 1e-4 parity by construction
        zero = ref == 0
        assert np.array_equal(got == 0, zero), name
        assert max_rel_err(got[~zero], ref[~zero]) <= 1e-4, name
        assert np.array_equal(feats.cpu().numpy(), np.stack([G.cpu().numpy().real, G.cpu().numpy().imag], -1))


def test_captured_graph_survives_a_larger_eager_call():
    """The forward's workspace is cached per stream and replaced when a larger shape arrives.  A graph captured on
    that stream must not be left pointing at memory the cache gave back: the captured call has a workspace of its own,
    and superseded buffers are kept.  Capture at a small shape after eager warm-up on the same stream, run a larger
    eager call there, fill the allocator with other tensors, replay."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import bkcore
    B, N = 12, 3 * 4096 + 7
    g = torch.Generator().manual_seed(15)
    a = (-2.0 + 0.5 * torch.randn(B, N, generator=g)).cuda()
    b = torch.ones(B, N - 1, device="cuda")
    z = torch.tensor(1j, dtype=torch.complex64, device="cuda")
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            want = bkcore.bk_forward(a, b, b, z)[1].clone()       # eager on s: the cached workspace now exists
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        _, Gg = bkcore.bk_forward(a, b, b, z)
    with torch.cuda.stream(s):
        big = (-2.0 + 0.5 * torch.randn(700, 6 * 4096, generator=g)).cuda()
        bkcore.bk_forward(big, torch.ones(700, 6 * 4096 - 1, device="cuda"), torch.ones(700, 6 * 4096 - 1, device="cuda"), z)
        junk = [torch.full((1 << 18,), float("nan"), device="cuda") for _ in range(64)]   # reuse whatever was freed
    torch.cuda.synchronize()
    for _ in range(3):
        Gg.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(torch.view_as_real(Gg), torch.view_as_real(want))
    del junk


def test_lazily_conjugated_z_and_feature_storage(golden):
    """z.conj() is a view with a flag, not new bits: the operator must compute with the conjugated value.  And the
    features share G_ii's storage by default (bk_core.py:296-298 builds a second tensor of the same numbers): an
    in-place edit of either output is refused by autograd instead of corrupting the saved G_ii, and
    SHARE_FEATURES_STORAGE = False gives independent tensors with the same gradients."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import bkcore
    cs = golden["z_python_complex"]
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    a, b, c = dev(cs["a"]), dev(cs["b"]), dev(cs["c"])
    zc = complex(cs["z"][0], cs["z"][1])
    z_conj_view = torch.tensor(zc.conjugate(), dtype=torch.complex64, device="cuda").conj()
    assert z_conj_view.is_conj()
    _, G = bkcore.BKCoreFunction.apply(a, b, c, z_conj_view)
    assert max_rel_err(G.cpu().numpy(), cs["G"]) <= 1e-4
    gf = dev(cs["gfeat"])
    grads = {}
    for share in (True, False):
        bkcore.BKCoreFunction.SHARE_FEATURES_STORAGE = share
        try:
            ar = a.clone().requires_grad_(True)
            feats, G = bkcore.BKCoreFunction.apply(ar, b, c, zc)
            assert (feats.data_ptr() == torch.view_as_real(G).data_ptr()) == share
            (feats * gf).sum().backward()
            grads[share] = ar.grad.clone()
            ar = a.clone().requires_grad_(True)
            feats, G = bkcore.BKCoreFunction.apply(ar, b, c, zc)
            if share:                                  # refused on the spot: the features are a view of the saved G_ii
                with pytest.raises(RuntimeError, match="modified inplace"):
                    feats.mul_(2.0)
                with pytest.raises(RuntimeError, match="inplace"):     # at the edit, or at backward (saved tensor)
                    G.mul_(2.0)
                    (feats * gf).sum().backward()
            else:
                feats.mul_(2.0)                        # independent storage: G_ii is intact
                (feats * gf).sum().backward()
                assert torch.equal(ar.grad, 2.0 * grads[False])
        finally:
            bkcore.BKCoreFunction.SHARE_FEATURES_STORAGE = True
    assert torch.equal(grads[True], grads[False])
