"""Pins the CPU oracle (oracle/bk_oracle_torch.py) to the reference implementation through the
golden fixtures that oracle/gen_golden.py recorded from it, and to the check the reference's
own test makes (dense inverse, tests/test_bk_core.py:47-77 of the reference)."""
import numpy as np
import pytest
import torch

from conftest import max_rel_err
from oracle import bk_oracle_torch as O

ALL = ["kat4", "lap_n1_b1", "lap_n1_b8", "lap_n2_b1", "lap_n2_b8", "lap_n3_b1", "lap_n3_b8", "lap_n8_b1",
       "lap_n8_b8", "lap_n64_b1", "lap_n64_b8", "lap_n257_b1", "lap_n257_b8", "lap_n1024_b8", "diag_only",
       "diag_only_bf16", "varying_bc", "varying_bc_small", "extreme_large", "extreme_small", "clamped_inputs",
       "poisoned", "nan_offdiag_only", "z_python_complex", "bench_randn", "f64_inputs", "big_diag", "tiny_diag_big_k"]


def _z(cs):
    return complex(cs["z"][0], cs["z"][1])


def _t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def test_fixture_list_complete(golden):
    assert sorted(golden.keys()) == sorted(ALL)


def test_kat4_values(golden):
    # the reference's own N=4 Laplacian case (tests/test_bk_core.py:21-45): value recorded from the reference
    G = golden["kat4"]["G"][0]
    assert abs(G[0] - (-0.37389147 + 0.30060643j)) < 1e-6
    assert abs(G[1] - (-0.32524389 + 0.40617782j)) < 1e-6
    assert abs(G[0] - G[3]) < 1e-7 and abs(G[1] - G[2]) < 1e-7


@pytest.mark.parametrize("name", ALL)
def test_forward_matches_reference(golden, name):
    cs = golden[name]
    feats, G = O.forward(_t(cs["a"]), _t(cs["b"]), _t(cs["c"]), _z(cs))
    # complex64 rounding of identical complex128 arithmetic: a few ulp
    nan_ref = np.isnan(cs["G"])
    assert np.array_equal(np.isnan(G.numpy()), nan_ref)
    assert max_rel_err(G.numpy()[~nan_ref], cs["G"][~nan_ref], 1e-30) < 1e-6
    assert feats.dtype == torch.float32 and G.dtype == torch.complex64
    # features are exactly (Re G, Im G) in both implementations (bk_core.py:296-298)
    assert np.array_equal(cs["features"][~nan_ref], np.stack([cs["G"].real, cs["G"].imag], -1)[~nan_ref])
    assert np.array_equal(feats.numpy()[~nan_ref], np.stack([G.numpy().real, G.numpy().imag], -1)[~nan_ref])
    if "G_raw" in cs:
        Gr = O.get_diag(_t(cs["a"]), _t(cs["b"]), _t(cs["c"]), _z(cs))
        assert max_rel_err(Gr.numpy(), cs["G_raw"], 1e-30) < 1e-6


@pytest.mark.parametrize("name", ALL)
@pytest.mark.parametrize("alpha", [0.0, 0.5, 1.0])
@pytest.mark.parametrize("with_gG", [False, True])
def test_reference_backward_matches(golden, name, alpha, with_gG):
    cs = golden[name]
    G = _t(cs["G"])
    ref = cs[f"grad_ref_a{alpha}_gG{int(with_gG)}"]
    got = O.backward_reference(G, _t(cs["gfeat"]), _t(cs["gG"]) if with_gG else None, alpha).numpy()
    np.testing.assert_allclose(got, ref, rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("name", [n for n in ALL if n not in ("poisoned", "nan_offdiag_only")])
def test_analytic_grads_match_reference_autograd(golden, name):
    cs = golden[name]
    a = _t(cs["a"]).double().clamp(-100, 100)
    b = _t(cs["b"]).double().clamp(-10, 10)
    c = _t(cs["c"]).double().clamp(-10, 10)
    ga, gb, gc = O.analytic_grads(a, b, c, _z(cs), grad_features=_t(cs["gfeat"]))
    for got, key in ((ga, "grad_true_a"), (gb, "grad_true_b"), (gc, "grad_true_c")):
        ref = cs[key]
        if ref.size == 0:
            assert got.numel() == 0
            continue
        scale = np.abs(ref).max()
        assert np.abs(got.numpy() - ref).max() <= 1e-7 * max(scale, 1e-30) + 1e-12


@pytest.mark.parametrize("name", ["lap_n8_b1", "lap_n8_b8", "lap_n64_b8", "kat4"])
def test_oracle_vs_dense_inverse(golden, name):
    # the reference's only value-level pin: max rel err vs torch.linalg.inv < 1e-3 (test_bk_core.py:74-77)
    cs = golden[name]
    G = O.get_diag(_t(cs["a"]), _t(cs["b"]), _t(cs["c"]), _z(cs)).numpy()
    rel = np.abs(G - cs["G_dense"]) / (np.abs(cs["G_dense"]) + 1e-6)
    assert rel.max() < 1e-3
    dense_here = O.dense_resolvent_diag(_t(cs["a"]), _t(cs["b"]), _t(cs["c"]), _z(cs)).numpy()
    assert max_rel_err(dense_here, cs["G_dense"]) < 1e-10


def test_exact_phi_convention_matches_dense_on_varying_bc(golden):
    # with position-varying b*c the reference convention is O(1) away from the resolvent; "exact" is not
    cs = golden["varying_bc_small"]
    a, b, c = _t(cs["a"]), _t(cs["b"]), _t(cs["c"])
    G_exact = O.get_diag(a, b, c, _z(cs), phi_k_index="exact").numpy()
    G_ref = O.get_diag(a, b, c, _z(cs), phi_k_index="reference").numpy()
    assert max_rel_err(G_exact, cs["G_dense"]) < 2e-3
    assert max_rel_err(G_ref, cs["G_raw"], 1e-30) < 1e-6
    assert max_rel_err(G_ref, cs["G_dense"]) > 1e-2


def test_batched_equals_per_row(golden):
    # mirror of the reference's batched-vs-loop test (tests/test_bk_core.py:79-109)
    cs = golden["lap_n64_b8"]
    a, b, c = _t(cs["a"]), _t(cs["b"]), _t(cs["c"])
    Gb = O.get_diag(a, b, c, 1j)
    rows = torch.cat([O.get_diag(a[i:i + 1], b[i:i + 1], c[i:i + 1], 1j) for i in range(a.shape[0])])
    assert torch.allclose(Gb, rows, rtol=1e-5, atol=1e-6)


def test_empty_sequence():
    G = O.get_diag(torch.zeros(3, 0), torch.zeros(3, 0), torch.zeros(3, 0), 1j)
    assert G.shape == (3, 0) and G.dtype == torch.complex64
