"""AbstractNFFTs convention (src/abstractNFFTs.jl:58-66,147-155,198-221): points in [-1/2, 1/2), opposite
sign of the exponent, increasing-frequency ordering (fftshift = true).  The reference's own test
(test/abstractNFFTs.jl:11-64, dims (512,) and (64, 81), Np = 1000) compares with NFFT.jl, which is
third-party and not available; the same definition is checked here against direct sums:
    adjoint:  f_hat[k] = sum_j f_j exp(+2 pi i k.x_j),   forward:  f_j = sum_k f_hat[k] exp(-2 pi i k.x_j),
    k_d = -N_d/2 ... ceil(N_d/2) - 1 in increasing order."""
import numpy as np
import pytest

from oracle import nufft_oracle as O


def _nfft_freqs(N):
    return np.arange(-(N // 2), -(N // 2) + N, dtype=np.float64)       # fftshift(fftfreq(N, N))


@pytest.mark.parametrize("dims", [(512,), (64, 81)])
def test_nfft_convention_against_direct_sums(dims):
    rng = np.random.default_rng(43)
    Np = 1000
    xp = [rng.random(Np) - 0.5 for _ in dims]
    vp = rng.standard_normal(Np) + 1j * rng.standard_normal(Np)
    plan = O.OraclePlan(dims, is_real=False, M=5, sigma=2.0, fftshift=True, point_transform=O.POINT_TRANSFORM_NFFT)
    O.set_points(plan, xp)
    us = O.exec_type1(plan, vp)                                          # adjoint(p) * vp
    ks = [_nfft_freqs(N) for N in dims]
    E = [np.exp(2j * np.pi * np.outer(ks[d], xp[d])) for d in range(len(dims))]
    exact = E[0] @ vp if len(dims) == 1 else np.einsum("bp,ap,p->ba", E[1], E[0], vp, optimize=True)
    assert O.l2_error(us, exact) < 1e-9                                  # reltol of the reference's test
    wp = O.exec_type2(plan, exact)                                       # p * us
    if len(dims) == 1:
        exact2 = np.conj(E[0]).T @ exact
    else:
        exact2 = np.einsum("bp,ap,ba->p", np.conj(E[1]), np.conj(E[0]), exact, optimize=True)
    assert O.l2_error(wp, exact2) < 1e-9


def test_point_convention_map():
    """src/abstractNFFTs.jl:147-155: x in [-1/2, 1/2) -> -2 pi x folded to [0, 2 pi)."""
    x = np.array([-0.5, -0.25, 0.0, 0.25, 0.499])
    t = O.nfft_point_convention(x)
    assert np.allclose(t, [np.pi, np.pi / 2, 0.0, 2 * np.pi - np.pi / 2, 2 * np.pi - 0.998 * np.pi])
    assert np.all((t >= 0) & (t < 2 * np.pi))
    assert O.nfft_point_convention(np.float32([0.1])).dtype == np.float32
