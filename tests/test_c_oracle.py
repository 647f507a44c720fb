"""The plain-C blocked-CPU restatement (oracle/nufft_oracle.c) against the numpy oracle."""
import numpy as np
import pytest

from oracle import nufft_oracle as O
from oracle import c_oracle as CO

pytestmark = pytest.mark.skipif(not CO.available(), reason="oracle/libnufft_oracle.so not built (make -C oracle)")


@pytest.mark.parametrize("dims,is_real,M,sigma,mode", [
    ((24, 20, 16), True, 4, 2.0, O.FAST_APPROXIMATION),
    ((24, 20, 16), False, 6, 1.5, O.DIRECT),
    ((35, 64, 40), True, 4, 1.5, O.DIRECT),
    ((64, 48), True, 4, 2.0, O.DIRECT),
    ((37, 41), False, 5, 1.25, O.FAST_APPROXIMATION),
    ((256,), False, 8, 2.0, O.FAST_APPROXIMATION),
    ((256,), True, 4, 2.0, O.DIRECT),
])
def test_c_oracle_matches_numpy_oracle(dims, is_real, M, sigma, mode):
    rng = np.random.default_rng(5)
    Np = 3000
    p = O.OraclePlan(dims, is_real=is_real, M=M, sigma=sigma, evalmode=mode)
    xs = [(rng.random(Np) * 3 - 1) * O.TWO_PI for _ in dims]
    v = rng.standard_normal(Np) if is_real else rng.standard_normal(Np) + 1j * rng.standard_normal(Np)
    O.set_points(p, xs)
    assert O.l2_error(CO.spread(p, [v])[0], O.spread(p, [v])[0]) < 1e-13
    assert O.l2_error(CO.exec_type1(p, v), O.exec_type1(p, v)) < 1e-13
    shape = tuple(reversed(p.size))
    uh = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
    assert O.l2_error(CO.exec_type2(p, uh), O.exec_type2(p, uh)) < 1e-13


def test_c_oracle_medium_3d_against_exact_modes():
    """A size numpy's add.at cannot reach comfortably: 64^3, 2e5 points; spot-check 40 modes against the
    exact sum (error ceiling of test/accuracy.jl at m = 4, sigma = 2)."""
    rng = np.random.default_rng(9)
    dims, Np = (64, 64, 64), 200_000
    p = O.OraclePlan(dims, is_real=True, M=4, sigma=2.0, evalmode=O.FAST_APPROXIMATION)
    xs = [rng.random(Np) * O.TWO_PI for _ in dims]
    v = rng.standard_normal(Np)
    O.set_points(p, xs)
    u = CO.exec_type1(p, v)
    idx = [tuple(rng.integers(0, n) for n in u.shape) for _ in range(40)]
    num = den = 0.0
    for (i3, i2, i1) in idx:
        k = (p.ks[0][i1], p.ks[1][i2], p.ks[2][i3])
        exact = np.sum(v * np.exp(-1j * (k[0] * xs[0] + k[1] * xs[1] + k[2] * xs[2])))
        num += abs(u[i3, i2, i1] - exact) ** 2
        den += abs(exact) ** 2
    assert np.sqrt(num / den) < 3 * 6 * 10.0 ** (-1.9 * 4)
