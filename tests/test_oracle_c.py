"""Pins the C oracle (oracle/bk_oracle.c) to the golden fixtures recorded from the reference."""
import numpy as np
import pytest

from conftest import max_rel_err
from oracle import bk_oracle_c as C
from test_oracle_golden import ALL


def _z(cs):
    return complex(cs["z"][0], cs["z"][1])


@pytest.mark.parametrize("name", ALL)
def test_c_forward_matches_reference(golden, name):
    cs = golden[name]
    if cs["a"].dtype == np.float64:
        pytest.skip("float32 entry point")
    F, G = C.forward_f32(cs["a"], cs["b"], cs["c"], _z(cs))
    nan_ref = np.isnan(cs["G"])
    assert np.array_equal(np.isnan(G), nan_ref)
    assert max_rel_err(G[~nan_ref], cs["G"][~nan_ref], 1e-30) < 1e-6
    assert np.array_equal(F[~nan_ref], np.stack([G.real, G.imag], -1)[~nan_ref])


@pytest.mark.parametrize("name", [n for n in ALL if n not in ("poisoned", "nan_offdiag_only")])
def test_c_get_diag_matches_reference(golden, name):
    cs = golden[name]
    G = C.get_diag(cs["a"], cs["b"], cs["c"], _z(cs))
    assert max_rel_err(G, cs["G_raw"], 1e-30) < 1e-6


@pytest.mark.parametrize("name", ALL)
@pytest.mark.parametrize("alpha", [0.0, 0.5, 1.0])
@pytest.mark.parametrize("with_gG", [False, True])
def test_c_backward_ref(golden, name, alpha, with_gG):
    cs = golden[name]
    got = C.backward_ref(cs["G"], cs["gfeat"], cs["gG"] if with_gG else None, alpha)
    np.testing.assert_allclose(got, cs[f"grad_ref_a{alpha}_gG{int(with_gG)}"], rtol=2e-5, atol=1e-6)


def test_c_exact_mode_vs_dense(golden):
    cs = golden["varying_bc_small"]
    G = C.get_diag(cs["a"], cs["b"], cs["c"], _z(cs), phi_exact=True)
    assert max_rel_err(G, cs["G_dense"]) < 2e-3


def test_c_broadcast_and_empty():
    a = np.zeros((3, 0))
    G = C.get_diag(a, a, a, 1j)
    assert G.shape == (3, 0)
