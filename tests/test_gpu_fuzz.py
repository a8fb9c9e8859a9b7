"""Random shapes, operand types, phi conventions and z through the Python mirror against the oracles on row subsets:
forward against the C oracle (1e-4), reference-mode backward against the oracle's formula (1e-4 of the size of its
terms), analytic backward with and without kept boundary states against the float64 adjoint (1e-4 of the largest
entry).  A row in which the float32 forward and the oracle sit on different sides of the clip at |G| = 50
(bk_core.py:204-208) is left out of the gradient comparison: the clip's derivative jumps there."""
import numpy as np
import pytest
import torch

from conftest import max_rel_err

pytestmark = pytest.mark.gpu


def _draw(rng):
    kind = rng.randint(0, 4)
    if kind == 0:
        B, N = int(rng.randint(1, 40)), int(rng.randint(1, 300))
    elif kind == 1:
        B, N = int(rng.randint(1, 1600)), int(rng.randint(4000, 40000))
    elif kind == 2:
        B, N = int(rng.randint(1, 12)), int(rng.randint(50000, 200000))
    else:
        B, N = int(rng.randint(100, 800)), int(rng.randint(8000, 50000))
    while B * N > (1 << 24):
        B = max(1, B // 2)
    dt = [torch.float32, torch.float32, torch.bfloat16, torch.float16, torch.float64][rng.randint(0, 5)]
    return B, N, dt, bool(rng.randint(0, 2)), bool(rng.randint(0, 2)), \
        complex(float(rng.uniform(-0.5, 0.5)), float(rng.uniform(0.8, 1.5))), int(rng.randint(0, 1 << 30))


@pytest.mark.parametrize("case", range(24))
def test_random_case_against_the_oracles(case):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import bkcore
    from oracle import bk_oracle_c as C
    from oracle.bk_analytic_adjoint import adjoint_grads
    rng = np.random.RandomState(7000 + case)
    B, N, dt, exact, varying, z, gseed = _draw(rng)
    g = torch.Generator().manual_seed(gseed)
    a = (-2.0 + 0.5 * torch.randn(B, N, generator=g)).to(dt)
    nb = max(N - 1, 0)
    amp = 0.1 if varying else 0.0
    b = (1.0 + amp * torch.randn(B, nb, generator=g)).to(dt)
    c = (1.0 + amp * torch.randn(B, nb, generator=g)).to(dt)
    gf = torch.randn(B, N, 2, generator=g)
    ad, bd, cd = a.cuda(), b.cuda(), c.cuda()
    feats, G = bkcore.bk_forward(ad, bd, cd, z, phi_exact=exact)
    rows = sorted(set(rng.randint(0, B, size=3).tolist()) | {0, B - 1})
    a64, b64, c64 = a[rows].double().numpy(), b[rows].double().numpy(), c[rows].double().numpy()
    ref = C.forward_f32(a64, b64, c64, z, phi_exact=exact)[1]
    Gh = G[rows].cpu().numpy()
    assert max_rel_err(Gh, ref) <= 1e-4, (B, N, dt, exact, varying, z)
    # reference-mode backward: the oracle's formula on the oracle's G, relative to the size of its terms
    grad = bkcore.bk_backward_reference(G, gf.cuda(), None, 0.5)
    gref = C.backward_ref(ref, gf[rows].numpy(), None, 0.5)
    m2 = np.abs(ref) ** 2
    gb = np.hypot(gf[rows, :, 0].numpy(), gf[rows, :, 1].numpy())
    terms = 0.5 * gb * (m2 + 1.0 / np.maximum(m2, 1e-4))
    on_clip = np.abs(np.abs(ref.astype(np.complex128)) - 50.0) < 1e-3     # (the formula clamps there: compare away from it)
    err = np.abs(grad[rows].cpu().numpy() - gref) / (terms + 1e-5)
    assert float(err[~on_clip].max(initial=0.0)) <= 1e-4, (B, N, dt, exact, varying, z)
    if dt != torch.float32 or N < 2 or N > 49152:
        return
    flips = (np.abs(np.abs(Gh.astype(np.complex128)) - 50.0) < 2e-5) != (np.abs(np.abs(ref.astype(np.complex128)) - 50.0) < 2e-5)
    keep = [i for i in range(len(rows)) if not flips[i].any()]
    _, G2, st = bkcore.bk_forward(ad, bd, cd, z, phi_exact=exact, want_states=True)
    assert st is not None
    gbar = (gf[rows, :, 0].double() + 1j * gf[rows, :, 1].double()).numpy()
    r3 = adjoint_grads(a64, b64, c64, z, gbar, "exact" if exact else "reference")[:3]
    for states in (None, st):
        ga = bkcore.bk_backward_analytic(ad, bd, cd, z, gf.cuda(), None, phi_exact=exact, G=G2, states=states)
        for x, r in zip(ga, r3):
            got = x[rows].cpu().numpy()
            for i in keep:
                assert np.abs(got[i] - r[i]).max() <= 1e-4 * np.abs(r[i]).max(), (B, N, exact, varying, z, states is not None)
