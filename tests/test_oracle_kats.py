"""Pins the CPU oracle against the reference's own known-answer tests.

The reference holds no golden-vector files: each of its tests computes the expectation on the fly
(direct NUDFT, plain FFT, exact cell indices) and asserts an error ceiling.  These tests re-create them
with the same sizes and parameters (own seeded RNG: Julia's Xoshiro(42) stream is not reproducible
without Julia) and assert the *same ceilings*.
"""
import numpy as np
import pytest

from oracle import nufft_oracle as O


def bkb_ceiling(T, M, sigma):
    """check_nufft_error for BackwardsKaiserBesselKernel, test/accuracy.jl:29-49."""
    if np.dtype(T) == np.float64:
        if abs(sigma - 1.25) < 1e-12:
            return max(10.0 ** (-1.20 * M), 4e-12)
        if abs(sigma - 2.0) < 1e-12:
            return max(6 * 10.0 ** (-1.9 * M), 4e-14)
    else:
        if abs(sigma - 1.25) < 1e-12:
            return 2 * 10.0 ** (-1.20 * M)
        if abs(sigma - 2.0) < 1e-12:
            return 6 * 10.0 ** (-1.9 * M)
    return None


def _points_1d(rng, T, Np):
    x = (rng.random(Np) * O.TWO_PI).astype(T)
    x = x + rng.integers(-1, 2, Np).astype(T) * T(O.TWO_PI)      # test/accuracy.jl:114-117
    return x.astype(T)


CASES_1D = [(np.float64, r, M, s) for r in (True, False) for M in range(4, 11) for s in (1.25, 2.0)] + \
           [(np.float32, r, 2, s) for r in (True, False) for s in (1.25, 2.0)]


@pytest.mark.parametrize("T,is_real,M,sigma", CASES_1D)
@pytest.mark.parametrize("evalmode", [O.DIRECT, O.FAST_APPROXIMATION])
def test_accuracy_1d_type1_type2(T, is_real, M, sigma, evalmode):
    """test/accuracy.jl:91-250: N = 256, Np = 512, type-1 and type-2 vs the exact sums."""
    N, Np = 256, 512
    rng = np.random.default_rng(42)
    plan = O.OraclePlan((N,), is_real=is_real, dtype=T, M=M, sigma=sigma, evalmode=evalmode)
    x = _points_1d(rng, T, Np)
    v = rng.standard_normal(Np).astype(T) if is_real else (rng.standard_normal(Np) + 1j * rng.standard_normal(Np)).astype(plan.cdtype)
    O.set_points(plan, [x])
    ceiling = bkb_ceiling(T, M, sigma)
    err1 = O.l2_error(O.exec_type1(plan, v), O.nudft_type1(plan.ks, [x], v))
    assert err1 < ceiling
    uh = (rng.standard_normal(len(plan.ks[0])) + 1j * rng.standard_normal(len(plan.ks[0]))).astype(plan.cdtype)
    exact = O.nudft_type2_real(plan, [x], uh) if is_real else O.nudft_type2(plan.ks, [x], uh)
    err2 = O.l2_error(O.exec_type2(plan, uh), exact)
    assert err2 < ceiling


@pytest.mark.parametrize("is_real", [True, False])
@pytest.mark.parametrize("M", [4, 5, 6, 7, 8])
def test_accuracy_2d(is_real, M):
    """test/multidimensional.jl:139-159: 64^2, sigma = 1.25, M = 4..8, ceiling of the 1-D test."""
    Ns, Np, sigma = (64, 64), 1000, 1.25
    rng = np.random.default_rng(M)
    plan = O.OraclePlan(Ns, is_real=is_real, M=M, sigma=sigma)
    xs = [_points_1d(rng, np.float64, Np) for _ in Ns]
    v = rng.standard_normal(Np) if is_real else rng.standard_normal(Np) + 1j * rng.standard_normal(Np)
    O.set_points(plan, xs)
    ceiling = 2 * bkb_ceiling(np.float64, M, sigma)      # two dimensions: errors of both directions add up
    assert O.l2_error(O.exec_type1(plan, v), O.nudft_type1(plan.ks, xs, v)) < ceiling
    shape = tuple(reversed(plan.size))
    uh = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
    exact = O.nudft_type2_real(plan, xs, uh) if is_real else O.nudft_type2(plan.ks, xs, uh)
    assert O.l2_error(O.exec_type2(plan, uh), exact) < ceiling


def test_grid_not_multiple_of_block_sizes():
    """test/multidimensional.jl:171-180: 37^2 with sigma = 2 (real: 80 x 75 oversampled)."""
    plan = O.OraclePlan((37, 37), is_real=True, M=4, sigma=2.0)
    assert plan.Nover == (80, 75)
    rng = np.random.default_rng(0)
    xs = [rng.random(400) * O.TWO_PI for _ in range(2)]
    v = rng.standard_normal(400)
    O.set_points(plan, xs)
    assert O.l2_error(O.exec_type1(plan, v), O.nudft_type1(plan.ks, xs, v)) < 2 * bkb_ceiling(np.float64, 4, 2.0)


@pytest.mark.parametrize("T", [np.float32, np.float64])
def test_point_to_cell_near_2pi(T):
    """test/near_2pi.jl:19-46: (x / L) * N never leaves the last cell for x = prevfloat(2π)."""
    L = T(2) * T(np.pi)
    x = np.nextafter(L, T(0))
    for N in range(400, 10001, 100):
        i, _ = O.point_to_cell(np.array([x], dtype=T), N)
        assert int(i[0]) == N - 1


def test_point_to_cell_thirds_and_pi():
    """test/near_2pi.jl:72-85,97-102."""
    L = O.TWO_PI
    for k, expect in ((1, 0), (2, 1), (3, 2)):
        x = np.nextafter(k * L / 3, 0.0)
        assert int(O.point_to_cell(np.array([x]), 3)[0][0]) == expect
    x = np.nextafter(np.pi, 0.0)
    i, _ = O.point_to_cell(np.array([x]), 24)
    dx = L / 24
    assert i[0] * dx <= x < (i[0] + 1) * dx


def test_single_point_near_2pi_and_pi():
    """test/near_2pi.jl:48-70 (M = 8, sigma = 1.5, rtol 1e-11) and :104-113 (M = 4, rtol 1e-5)."""
    N = 32
    plan = O.OraclePlan((N,), is_real=False, M=8, sigma=1.5)
    x = np.array([np.nextafter(O.TWO_PI, 0.0)])
    v = np.array([4.2 + 3j])
    O.set_points(plan, [x])
    u = O.exec_type1(plan, v)
    exact = O.nudft_type1(plan.ks, [x], v)
    assert np.linalg.norm(u - exact) <= 1e-11 * np.linalg.norm(exact)
    plan = O.OraclePlan((16,), is_real=True, M=4, sigma=1.5)
    x = np.array([np.nextafter(np.pi, 0.0)])
    v = np.array([3.4])
    O.set_points(plan, [x])
    u = O.exec_type1(plan, v)
    exact = O.nudft_type1(plan.ks, [x], v)
    assert np.linalg.norm(u - exact) <= 1e-5 * np.linalg.norm(exact)


@pytest.mark.parametrize("is_real", [True, False])
def test_uniform_points_equal_plain_fft(is_real):
    """test/uniform_points.jl:17-61: equispaced points => type-1 == fft/rfft (err < 4e-10),
    type-2 == bfft/brfft (err < 5e-10); N = 256, M = 8, sigma = 1.25."""
    N, M, sigma = 256, 8, 1.25
    plan = O.OraclePlan((N,), is_real=is_real, M=M, sigma=sigma)
    x = np.arange(N) * (O.TWO_PI / N)
    rng = np.random.default_rng(1)
    v = rng.standard_normal(N) if is_real else rng.standard_normal(N) + 1j * rng.standard_normal(N)
    if is_real:      # zero-out the Nyquist mode "to avoid comparison issues" (test/uniform_points.jl:24-27)
        r = np.fft.rfft(v)
        r[-1] = 0
        v = np.fft.irfft(r, n=N)
    O.set_points(plan, [x])
    u = O.exec_type1(plan, v)
    ref = np.fft.rfft(v) if is_real else np.fft.fft(v)
    if is_real:
        ref[-1] = 0
    assert O.l2_error(u, ref) < 4e-10
    w = O.exec_type2(plan, ref.astype(np.complex128))
    back = np.fft.irfft(ref, n=N) * N if is_real else np.fft.ifft(ref) * N
    assert O.l2_error(w, back) < 5e-10


def test_polynomial_window_matches_direct_window():
    """test/approx_window_functions.jl:9-24: 1000 x values, same cell, values agree to rtol 1e-7
    (M = 4, sigma = 1.5, N = 256)."""
    pd = O.OraclePlan((256,), is_real=False, M=4, sigma=1.5, evalmode=O.DIRECT)
    pf = O.OraclePlan((256,), is_real=False, M=4, sigma=1.5, evalmode=O.FAST_APPROXIMATION)
    dx = O.TWO_PI / pd.Nover[0]
    x = np.concatenate([np.linspace(0.8, 2.2, 1000) * dx, np.linspace(0.0, O.TWO_PI, 1000, endpoint=False)])
    i0, v0 = O.evaluate_window(pd, 0, x)
    i1, v1 = O.evaluate_window(pf, 0, x)
    assert np.array_equal(i0, i1)                                       # same bin
    # `SVector(a.values) ≈ SVector(b.values) rtol=1e-7` is a norm-wise comparison of the 2M values
    rel = np.linalg.norm(v1 - v0, axis=1) / np.linalg.norm(v0, axis=1)
    assert rel.max() < 1e-7


def test_size_too_small_is_an_argument_error():
    """test/errors.jl:5-10."""
    with pytest.raises(ValueError):
        O.OraclePlan((4,), M=8, sigma=1.25)


def test_oversampled_size_rule():
    """src/plan.jl:485-498 on the BASELINE configurations and the survey's examples."""
    assert O.OraclePlan((256, 256, 256), is_real=True).Nover == (512, 512, 512)
    assert O.OraclePlan((256, 256, 256), is_real=True, sigma=1.5).Nover == (384, 384, 384)
    assert O.OraclePlan((512, 512, 512), is_real=False, M=8).Nover == (1024, 1024, 1024)
    assert O.OraclePlan((35, 64, 40), is_real=True, sigma=1.5).Nover == (54, 96, 60)
    assert O.nextprod235(37) == 40 and O.nextprod235(74) == 75 and O.nextprod235(1) == 1


def test_reference_lds_rule_reproduces_survey_numbers():
    """block_dims_gpu_shmem at 64 KiB (src/gpu_common.jl:19-92): F64 M=4 -> n = 12, Np = 46;
    CF64 M=4 -> n = 8; CF32 M=8 -> n = 4."""
    assert O.block_dims_gpu_shmem(8, 8, 3, 4, 16)[0] == 12
    assert O.block_dims_gpu_shmem(16, 8, 3, 4, 16)[0] == 8
    assert O.block_dims_gpu_shmem(8, 4, 3, 8, 16)[0] == 4
