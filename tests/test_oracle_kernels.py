"""Known-answer tests of the other spreading kernels (KaiserBessel, Gaussian, BSpline),
mirroring the reference's own tests for them with the same sizes, parameters and error ceilings:
test/accuracy.jl:7-88 (check_nufft_error per kernel), :252-300 (kernel x M x sigma loops, explicit
kernel parameter), test/approx_window_functions.jl:9-39 (Direct vs FastApproximation windows)."""
import math

import numpy as np
import pytest

from oracle import nufft_oracle as O

KERNELS = {"kb": O.KERNEL_KB, "gaussian": O.KERNEL_GAUSSIAN, "bspline": O.KERNEL_BSPLINE, "bkb": O.KERNEL_BKB}


def ceiling(kernel, T, M, sigma):
    """check_nufft_error, test/accuracy.jl:7-88 (None: the reference asserts nothing there)."""
    f64 = np.dtype(T) == np.float64
    if kernel == O.KERNEL_KB:
        if abs(sigma - 1.25) < 1e-12:
            return max(10.0 ** (-1.16 * M) * 1.05, 4e-12) if f64 else 2 * 10.0 ** (-1.16 * M)
        return max(6 * 10.0 ** (-1.9 * M), 4e-14) if f64 else 6 * 10.0 ** (-1.9 * M)
    if kernel == O.KERNEL_GAUSSIAN:
        return 10.0 ** (-0.95 * M) * 0.8 if abs(sigma - 2.0) < 1e-12 else None
    if kernel == O.KERNEL_BSPLINE:
        return 10.0 ** (-0.98 * M) * 0.4 if abs(sigma - 2.0) < 1e-12 else None
    raise AssertionError


def _points_1d(rng, T, Np):
    x = (rng.random(Np) * O.TWO_PI).astype(T)
    return (x + rng.integers(-1, 2, Np).astype(T) * T(O.TWO_PI)).astype(T)      # test/accuracy.jl:114-117


CASES = [(k, np.float64, r, M, s) for k, sigmas in (("kb", (1.25, 2.0)), ("gaussian", (2.0,)), ("bspline", (2.0,)))
         for r in (True, False) for M in range(4, 11) for s in sigmas] + \
        [(k, np.float32, r, 2, s) for k, sigmas in (("kb", (1.25, 2.0)), ("gaussian", (2.0,)), ("bspline", (2.0,)))
         for r in (True, False) for s in sigmas]


@pytest.mark.parametrize("kname,T,is_real,M,sigma", CASES)
@pytest.mark.parametrize("evalmode", [O.DIRECT, O.FAST_APPROXIMATION])
def test_accuracy_1d_other_kernels(kname, T, is_real, M, sigma, evalmode):
    """test/accuracy.jl:252-283: N = 256, Np = 512, type 1 and type 2 against the exact sums."""
    kernel = KERNELS[kname]
    N, Np = 256, 512
    rng = np.random.default_rng(1)     # own seed (Julia's Xoshiro(42) stream is not reproducible here)
    plan = O.OraclePlan((N,), is_real=is_real, dtype=T, M=M, sigma=sigma, evalmode=evalmode, kernel=kernel)
    x = _points_1d(rng, T, Np)
    v = rng.standard_normal(Np).astype(T) if is_real else (rng.standard_normal(Np) + 1j * rng.standard_normal(Np)).astype(plan.cdtype)
    O.set_points(plan, [x])
    c = ceiling(kernel, T, M, sigma)
    assert O.l2_error(O.exec_type1(plan, v), O.nudft_type1(plan.ks, [x], v)) < c
    uh = (rng.standard_normal(len(plan.ks[0])) + 1j * rng.standard_normal(len(plan.ks[0]))).astype(plan.cdtype)
    exact = O.nudft_type2_real(plan, [x], uh) if is_real else O.nudft_type2(plan.ks, [x], uh)
    assert O.l2_error(O.exec_type2(plan, uh), exact) < c


@pytest.mark.parametrize("kname", ["kb", "bkb"])
def test_explicit_kernel_parameter(kname):
    """test/accuracy.jl:285-297 "Setting kernel parameter": M = 2, sigma = 2, beta = M pi (2 - 1/sigma)
    passed explicitly; same ceiling as the default parameter."""
    M, sigma, N, Np = 2, 2.0, 256, 512
    beta = M * math.pi * (2 - 1 / sigma)
    for T, is_real in ((np.float64, True), (np.float64, False), (np.float32, True)):
        rng = np.random.default_rng(42)
        plan = O.OraclePlan((N,), is_real=is_real, dtype=T, M=M, sigma=sigma, kernel=KERNELS[kname], kernel_param=beta)
        assert abs(plan.betas[0] - float(np.dtype(T).type(beta))) == 0
        x = _points_1d(rng, T, Np)
        v = rng.standard_normal(Np).astype(T) if is_real else (rng.standard_normal(Np) + 1j * rng.standard_normal(Np)).astype(plan.cdtype)
        O.set_points(plan, [x])
        c = 6 * 10.0 ** (-1.9 * M)
        assert O.l2_error(O.exec_type1(plan, v), O.nudft_type1(plan.ks, [x], v)) < c


@pytest.mark.parametrize("kname", ["bspline", "gaussian", "kb", "bkb"])
def test_direct_and_fast_windows_agree(kname):
    """test/approx_window_functions.jl:9-39: sigma = 1.5, m = 4, N = 256, 1000 points in
    [0.8, 2.2] dx: same cell and values within rtol 1e-7 (norm-wise over the 2M values)."""
    kw = dict(is_real=False, M=4, sigma=1.5, kernel=KERNELS[kname])
    pd = O.OraclePlan((256,), evalmode=O.DIRECT, **kw)
    pf = O.OraclePlan((256,), evalmode=O.FAST_APPROXIMATION, **kw)
    dx = O.TWO_PI / 256          # the reference builds the kernel data on N = 256 grid points
    x = np.concatenate([np.linspace(0.8, 2.2, 1000) * dx, np.linspace(0.0, O.TWO_PI, 997, endpoint=False)])
    i0, v0 = O.evaluate_window(pd, 0, x)
    i1, v1 = O.evaluate_window(pf, 0, x)
    assert np.array_equal(i0, i1)
    rel = np.linalg.norm(v1 - v0, axis=1) / np.linalg.norm(v0, axis=1)
    assert rel.max() < 1e-7


def test_bspline_window_properties():
    """B-splines of order 2M on unit knots: partition of unity, non-negative, and the M = 1 case is the
    linear hat (src/Kernels/bspline.jl:131-139 describes the output order)."""
    x = np.linspace(1e-6, 1.0, 101)
    for M in (1, 2, 4, 7, 10):
        b = O.bspline_evaluate_all(x, 2 * M)
        assert b.shape == (101, 2 * M)
        assert np.all(b >= 0) and np.allclose(b.sum(axis=1), 1.0, atol=1e-14)
    b1 = O.bspline_evaluate_all(x, 2)
    assert np.allclose(b1[:, 0], x) and np.allclose(b1[:, 1], 1 - x)


def test_kernel_fourier_transforms_against_quadrature():
    """phi_hat(k) = integral of phi(x) exp(-ikx) over the support, checked by quadrature for each kernel
    (the analytic forms are src/Kernels/kaiser_bessel.jl:167-174, gaussian.jl:118-123 (untruncated
    Gaussian), bspline.jl:121-129)."""
    N, M = 64, 4
    dx = O.TWO_PI / N
    w = M * dx
    ks = np.array([0.0, 1.0, 5.0, 13.0])
    xs = np.linspace(-w, w, 200001)
    beta = O.kb_beta(M, 2.0)
    quad = np.array([np.trapezoid(O.kb_function(xs / w, beta) * np.cos(k * xs), xs) for k in ks])
    assert np.allclose(quad, O.kb_fourier(ks, M, N, beta), rtol=1e-8)
    # B-spline of order 2M: window values at x' in (0, 1] for every cell offset
    t = np.linspace(1e-9, 1.0, 20001)
    b = O.bspline_evaluate_all(t, 2 * M)                 # b[:, j]: node offset j - M + (1 - x')... symmetric support
    total = sum(np.trapezoid(b[:, j], t) for j in range(2 * M)) * dx
    assert abs(total - O.bspline_fourier(np.array([0.0]), M, N)[0]) < 1e-7 * dx
    tau = O.gaussian_tau(O.gaussian_ell(M, 2.0), N)
    xs = np.linspace(-12 * math.sqrt(tau), 12 * math.sqrt(tau), 400001)
    quad = np.array([np.trapezoid(np.exp(-xs * xs / tau) * np.cos(k * xs), xs) for k in ks])
    assert np.allclose(quad, O.gaussian_fourier(ks, tau), rtol=1e-9)


@pytest.mark.parametrize("M", [2, 3, 4, 6, 8, 10])
def test_float32_coefficients_float64_solve_vs_reference_float32_solve(M):
    """Stated deviation: the reference solves the Chebyshev-node Vandermonde system of the piecewise
    polynomial in T = real(Z) (`Matrix{T}`, src/Kernels/piecewise_polynomial.jl:50-60); the library (and the oracle's
    default) solve in Float64 and round the coefficients to Float32.  Pin the size of the difference: window values from
    the two coefficient sets agree to 3e-7 of the window maximum (the Float32 solve is the *less* accurate of the two
    against the exactly solved polynomial), i.e. far inside the reference's Float32 GPU-vs-CPU bound of 1e-5
    (test/pseudo_gpu.jl:159-171)."""
    beta = float(np.float32(O.bkb_beta(M, 2.0)))
    f = lambda y: O.bkb_function(y, beta)                       # noqa: E731
    c64 = O.piecewise_poly_coefficients(f, M, M + 4)
    c32 = O.piecewise_poly_coefficients(f, M, M + 4, solve_dtype=np.float32)
    X = np.linspace(0.0, 1.0, 257)
    xx = 2 * X - 1

    def horner(cs):
        v = np.broadcast_to(cs[-1][None, :], (len(X), 2 * M)).copy()
        for k in range(cs.shape[0] - 2, -1, -1):
            v = xx[:, None] * v + cs[k][None, :]
        return v

    exact = horner(c64)
    lib = horner(c64.astype(np.float32).astype(np.float64))     # what the library loads into Float32 plans
    ref = horner(c32.astype(np.float32).astype(np.float64))     # what the reference computes
    peak = np.abs(exact).max()
    assert np.abs(lib - ref).max() / peak < 3e-7
    assert np.abs(lib - exact).max() <= np.abs(ref - exact).max() * 1.5 + 1e-8 * peak
