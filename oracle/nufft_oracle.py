"""CPU oracle for the NUFFT hot path of jipolanco/NonuniformFFTs.jl (numpy restatement).

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import it.  The shipped path
(``nonuniformffts.jl_amd``) never touches this module and fails loudly when its HIP library
is missing.

Pinning status.  The reference is 100 % Julia and there is no Julia runtime in the build
container, so reference-generated outputs do not exist.  The reference's own test-suite holds
no golden-vector files either: every test there is a *known-answer* test (direct O(N*Np) NUDFT
under an error ceiling, FFT equivalence for equispaced points, cell-index edge cases,
polynomial-vs-direct window agreement).  This oracle is pinned against those known-answer
tests, re-created with the same sizes/parameters in ``tests/test_oracle_*.py``:
test/accuracy.jl:29-38,91-250, test/multidimensional.jl:139-180, test/near_2pi.jl:19-113,
test/uniform_points.jl:17-61, test/approx_window_functions.jl:9-24, test/errors.jl:5-10.
Parity with the running Julia package is therefore pinned through shared analytic
expectations, not through reference-generated vectors.

Every function cites the reference file:line (relative to /root/reference) that it restates.
Arrays follow the reference's column-major convention by storing grids with *reversed* axes:
a Julia array of size (N1, N2, N3) is a C-contiguous numpy array of shape (N3, N2, N1), so
that dimension 1 is the fastest one in memory exactly as in Julia.

Third-party arithmetic that is not under /root/reference: Bessels.jl ``besseli0`` (compat 0.2,
used for the kernel's Fourier transform, src/Kernels/kaiser_bessel_backwards.jl:143) is
replaced by ``scipy.special.i0``; FFTW by ``numpy.fft`` (pocketfft); ``LinearAlgebra.lu!`` by
``numpy.linalg.solve``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Optional, Sequence, Tuple

import numpy as np
from scipy.special import i0 as _bessel_i0

TWO_PI = 2.0 * math.pi

DIRECT = 0
FAST_APPROXIMATION = 1

# spreading kernels (values of NUFFT_KERNEL_* in include/nufft_mi355x.h)
KERNEL_BKB = 0          # BackwardsKaiserBesselKernel (default), src/Kernels/kaiser_bessel_backwards.jl
KERNEL_KB = 1           # KaiserBesselKernel, src/Kernels/kaiser_bessel.jl
KERNEL_GAUSSIAN = 2     # GaussianKernel, src/Kernels/gaussian.jl
KERNEL_BSPLINE = 3      # BSplineKernel, src/Kernels/bspline.jl

# point_transform (values of NUFFT_POINT_TRANSFORM_*)
POINT_TRANSFORM_IDENTITY = 0
POINT_TRANSFORM_NFFT = 1    # _transform_point_convention, src/abstractNFFTs.jl:147-155


# --------------------------------------------------------------------------------------
# Plan-time parameter math
# --------------------------------------------------------------------------------------

def nextprod235(n: int) -> int:
    """Smallest 2^a 3^b 5^c >= n (Julia ``nextprod((2, 3, 5), n)``, used at src/plan.jl:492-494)."""
    n = max(int(n), 1)
    best = None
    p2 = 1
    while True:
        p23 = p2
        while True:
            p = p23
            while p < n:
                p *= 5
            if best is None or p < best:
                best = p
            if p23 >= n:
                break
            p23 *= 3
        if p2 >= n:
            break
        p2 *= 2
    return best


def oversampled_size(N: int, sigma: float, real_first_dim: bool) -> int:
    """Oversampled grid size along one dimension (src/plan.jl:485-498)."""
    if real_first_dim:
        return 2 * nextprod235(int(math.floor(sigma * ((N + 1) // 2))))
    return nextprod235(int(math.floor(sigma * N)))


def fftfreq_int(N: int) -> np.ndarray:
    """AbstractFFTs.fftfreq(N, N): 0, 1, ..., ceil(N/2)-1, -floor(N/2), ..., -1."""
    k = np.arange(N)
    k[k >= (N + 1) // 2] -= N
    return k.astype(np.float64)


def rfftfreq_int(N: int) -> np.ndarray:
    """AbstractFFTs.rfftfreq(N, N): 0, 1, ..., N/2 (length N//2 + 1)."""
    return np.arange(N // 2 + 1, dtype=np.float64)


def init_wavenumbers(Ns: Sequence[int], is_real: bool):
    """src/plan.jl:558-566.  Returns one wavenumber vector per dimension (dimension 1 first)."""
    ks = []
    for d, N in enumerate(Ns):
        if is_real and d == 0:
            ks.append(rfftfreq_int(N))
        else:
            ks.append(fftfreq_int(N))
    return ks


def bkb_beta(M: int, sigma_d: float) -> float:
    """Shape parameter of the backwards Kaiser-Bessel kernel
    (src/Kernels/kaiser_bessel_backwards.jl:123-136)."""
    a = M * (2.0 - 1.0 / sigma_d)
    gamma = max(0.995, math.sqrt(1.0 - 0.3 / (a * a)))
    return math.pi * a * gamma


def bkb_function(y, beta):
    """phi(y) = sinh(beta sqrt(1 - y^2)) / (pi sqrt(1 - y^2)); value beta/pi at |y| = 1
    (src/Kernels/kaiser_bessel_backwards.jl:99-102,158-175)."""
    y = np.asarray(y, dtype=np.float64)
    z = 1.0 - y * y
    s = np.sqrt(np.maximum(z, 0.0))
    bs = beta * s
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = np.where(bs == 0.0, 1.0, np.sinh(bs) / np.where(bs == 0.0, 1.0, bs))
    return ratio * (beta / math.pi)


def piecewise_poly_coefficients(f, M: int, npoly: int, solve_dtype=np.float64) -> np.ndarray:
    """Chebyshev-node fit of ``f`` on each of the 2M sub-intervals of [-1, 1], ordered from
    right (+1) to left (-1) (src/Kernels/piecewise_polynomial.jl:23-74).

    Returns ``cs`` of shape (npoly, 2M): ``cs[k, j]`` multiplies x^k on sub-interval j
    (the "transposed" tuple-of-tuples layout of :43-47,73).

    ``solve_dtype``: the reference builds and LU-solves the Vandermonde system in T = real(Z) (`Matrix{T}`, :52-57).  The
    library and the default here solve in Float64 and round the coefficients to T afterwards — identical for Float64
    plans; for Float32 plans the Float32 solve (solve_dtype = float32: LAPACK sgesv, the routine family `lu!` / `ldiv!`
    call) loses 3-4 digits to the conditioning of the system, the Float64 solve does not.  DESIGN.md section 2 states the
    deviation; tests/test_oracle_kernels.py pins its size.
    """
    L = 2 * M
    st = np.dtype(solve_dtype).type
    i = np.arange(1, npoly + 1, dtype=np.float64)
    xs = np.cos(np.pi * (i - 0.5).astype(solve_dtype).astype(np.float64) / npoly).astype(solve_dtype)   # :61  cospi(T(i - 1/2) / N)
    A = np.empty((npoly, npoly), dtype=solve_dtype)   # :33-38  A[i, j] = xs[i]^(j-1), powers by repeated products in T
    pw = np.ones(npoly, dtype=solve_dtype)
    for jj in range(npoly):
        A[:, jj] = pw
        pw = (pw * xs).astype(solve_dtype)
    cs = np.empty((npoly, L), dtype=np.float64)
    delta = 1.0 / L                                   # :66
    for j in range(1, L + 1):
        h = 1.0 - 2.0 * (j - 0.5) / L                 # :65
        ys = np.asarray(f(h + xs.astype(np.float64) * delta)).astype(solve_dtype)   # :67-70 (h, δ are Float64 there too)
        cs[:, j - 1] = np.linalg.solve(A, ys)         # :39-40
    return cs


def bkb_poly_coefficients(M: int, beta: float) -> np.ndarray:
    """Polynomial degree M + 3 (Npoly = M + 4), src/Kernels/kaiser_bessel_backwards.jl:98-102."""
    return piecewise_poly_coefficients(lambda y: bkb_function(y, beta), M, M + 4)


def bkb_fourier(ks: np.ndarray, M: int, Nover: int, beta: float) -> np.ndarray:
    """phi_hat(k) = w I0(sqrt(beta^2 - (w k)^2)), w = M dx
    (src/Kernels/kaiser_bessel_backwards.jl:138-145, src/Kernels/Kernels.jl:108-117)."""
    w = M * (TWO_PI / Nover)
    q = w * np.asarray(ks, dtype=np.float64)
    s = np.sqrt(beta * beta - q * q)
    return w * _bessel_i0(s)


# ---- KaiserBesselKernel (src/Kernels/kaiser_bessel.jl) -----------------------------------------

def kb_beta(M: int, sigma_d: float) -> float:
    """beta = pi a gamma, a = M (2 - 1/sigma), gamma = sqrt(1 - 0.8 / a^2)
    (src/Kernels/kaiser_bessel.jl:151-165)."""
    a = M * (2.0 - 1.0 / sigma_d)
    gamma = math.sqrt(1.0 - 0.8 / (a * a))
    return math.pi * a * gamma


def kb_function(y, beta):
    """phi(y) = I0(beta sqrt(1 - y^2)) on |y| <= 1 (src/Kernels/kaiser_bessel.jl:128-130,197-210)."""
    y = np.asarray(y, dtype=np.float64)
    return _bessel_i0(beta * np.sqrt(np.maximum(1.0 - y * y, 0.0)))


def kb_poly_coefficients(M: int, beta: float) -> np.ndarray:
    """Npoly = M + 4, src/Kernels/kaiser_bessel.jl:127-130."""
    return piecewise_poly_coefficients(lambda y: kb_function(y, beta), M, M + 4)


def kb_fourier(ks: np.ndarray, M: int, Nover: int, beta: float) -> np.ndarray:
    """phi_hat(k) = 2 w sinh(s) / s, s = sqrt(beta^2 - (w k)^2), w = M dx
    (src/Kernels/kaiser_bessel.jl:167-174)."""
    w = M * (TWO_PI / Nover)
    q = w * np.asarray(ks, dtype=np.float64)
    s = np.sqrt(beta * beta - q * q)
    return 2.0 * w * np.sinh(s) / s


# ---- GaussianKernel (src/Kernels/gaussian.jl) --------------------------------------------------

def gaussian_ell(M: int, sigma_d: float) -> float:
    """ell / dx = sqrt(sigma M / (2 sigma - 1) / pi), Potts & Steidl eq. (5.9)
    (src/Kernels/gaussian.jl:107-116)."""
    return math.sqrt(sigma_d * M / (2.0 * sigma_d - 1.0) / math.pi)


def gaussian_tau(ell: float, Nover: int) -> float:
    """tau = 2 (ell dx)^2 (src/Kernels/gaussian.jl:76-80)."""
    sg = ell * (TWO_PI / Nover)
    return 2.0 * sg * sg


def gaussian_fourier(ks: np.ndarray, tau: float) -> np.ndarray:
    """phi_hat(k) = exp(-tau k^2 / 4) sqrt(pi tau) (src/Kernels/gaussian.jl:118-123)."""
    k = np.asarray(ks, dtype=np.float64)
    return np.exp(-tau * k * k / 4.0) * math.sqrt(math.pi * tau)


# ---- BSplineKernel (src/Kernels/bspline.jl) ----------------------------------------------------

def bspline_evaluate_all(x, k: int):
    """All k = 2M B-splines of order k that are non-zero at x in (0, 1], uniform unit knots, in the
    reference's output order (src/Kernels/bspline.jl:140-193: the generated recursion with
    alpha = 1 / (q - 1); ``bsplines_evaluate_step`` :176-193).  x: array (Np,) -> (Np, k)."""
    x = np.asarray(x)
    T = x.dtype.type
    bs = [np.ones_like(x)]
    for q in range(2, k + 1):
        alpha = T(1) / T(q - 1)
        ds = []
        xx = x.copy()
        for _ in range(q - 1):
            ds.append((alpha * xx).astype(x.dtype))
            xx = xx + T(1)
        new = [ds[0] * bs[0]]
        for j in range(2, q):
            new.append((T(1) - ds[j - 2]) * bs[j - 2] + ds[j - 1] * bs[j - 1])
        new.append((T(1) - ds[q - 2]) * bs[q - 2])
        bs = new
    return np.stack(bs, axis=1)


def bspline_fourier(ks: np.ndarray, M: int, Nover: int) -> np.ndarray:
    """phi_hat(k) = (sin(kh) / kh)^(2M) dt, kh = k dt / 2; dt at k = 0 (src/Kernels/bspline.jl:121-129)."""
    dt = TWO_PI / Nover
    k = np.asarray(ks, dtype=np.float64)
    kh = k * dt / 2.0
    with np.errstate(divide="ignore", invalid="ignore"):
        sinc = np.where(k == 0, 1.0, np.sin(kh) / np.where(k == 0, 1.0, kh))
    return np.where(k == 0, 1.0, sinc ** (2 * M)) * dt


def non_oversampled_indices(ks: np.ndarray, Nover_axis: int, fftshift: bool = False) -> np.ndarray:
    """0-based index map from the output wavenumbers to the oversampled axis
    (src/NonuniformFFTs.jl:318-348)."""
    Nk = len(ks)
    ax = np.arange(Nover_axis)
    indmap = np.empty(Nk, dtype=np.int64)
    r2c = ks[-1] > 0
    if r2c:
        indmap[:] = ax[:Nk]
    elif Nk % 2 == 0:
        h = Nk // 2
        if fftshift:
            indmap[:h] = ax[Nover_axis - h:]
            indmap[h:] = ax[:h]
        else:
            indmap[:h] = ax[:h]
            indmap[h:] = ax[Nover_axis - h:]
    else:
        h = (Nk - 1) // 2
        if fftshift:
            indmap[:h] = ax[Nover_axis - h:]
            indmap[h:] = ax[:h + 1]
        else:
            indmap[:h + 1] = ax[:h + 1]
            indmap[h + 1:] = ax[Nover_axis - h:] if h > 0 else ax[:0]
    return indmap


def block_dims_gpu_shmem(elt_bytes: int, real_bytes: int, D: int, M: int, np_min: int,
                         max_shmem: int = 64 << 10):
    """Reference LDS budgeting rule (src/gpu_common.jl:19-92) with the ROC extension's 64 KiB
    (ext/NonuniformFFTsAMDGPUExt.jl:65-70).  Returns (n, Np_actual)."""
    const_shmem = 8 * (2 + D) + 128
    per_point = real_bytes * D * 2 * M + 8 * D + elt_bytes
    max_local = (max_shmem - const_shmem - np_min * per_point) // elt_bytes
    m = int(math.floor(_invpow(max_local, D)))
    n = m - (2 * M - 1)
    left = max_shmem - const_shmem - elt_bytes * m ** D
    return n, left // per_point


def _invpow(x, D):
    if D == 1:
        return float(x)
    if D == 2:
        return math.sqrt(x)
    if D == 3:
        # Julia's cbrt is correctly rounded for perfect cubes; guard against 26.999999
        c = round(x ** (1.0 / 3.0))
        return float(c) if c ** 3 == x else x ** (1.0 / 3.0)
    return math.sqrt(math.sqrt(x))


@dataclass
class OraclePlan:
    """Host-side restatement of ``_PlanNUFFT`` (src/plan.jl:467-541).  ``kernel`` selects one of the
    four spreading kernels (default: BackwardsKaiserBesselKernel, the default kernel on CPU and ROC,
    src/NonuniformFFTs.jl:52, ext/NonuniformFFTsAMDGPUExt.jl:54); ``kernel_param`` overrides the shape
    parameter (beta for the two Kaiser-Bessel kernels, ell / dx for the Gaussian), as
    ``KaiserBesselKernel(beta)`` etc. do in the reference."""
    Ns: Tuple[int, ...]
    is_real: bool = True           # Z <: Real
    dtype: type = np.float64       # real(Z)
    M: int = 4
    sigma: float = 2.0
    evalmode: int = FAST_APPROXIMATION
    ntransforms: int = 1
    fftshift: bool = False
    kernel: int = KERNEL_BKB
    kernel_param: Optional[float] = None
    point_transform: int = POINT_TRANSFORM_IDENTITY
    # Precision of the coordinate arithmetic (fold, cell index, cell fraction) when it differs from `dtype`: a Float64
    # plan with coord_dtype = float32 takes Float32 points, locates them exactly as a Float32 plan of the reference does
    # (src/blocking/blocking.jl:26-33, src/Kernels/Kernels.jl:121-126 evaluated in T) and evaluates windows, sums and
    # FFTs in Float64 — the expectation for Float32 plans whose un-normalised window overflows Float32 (3-D, M >= 7).
    coord_dtype: Optional[type] = None
    # derived
    Nover: Tuple[int, ...] = field(init=False)
    ks: list = field(init=False)
    betas: list = field(init=False)      # shape parameter per dimension (beta; ell / dx for the Gaussian; 0 for B-splines)
    taus: list = field(init=False)       # Gaussian: tau = 2 (ell dx)^2
    coefs: list = field(init=False)
    phihat: list = field(init=False)
    index_map: list = field(init=False)
    points: Optional[list] = field(init=False, default=None)

    def __post_init__(self):
        self.Ns = tuple(int(n) for n in self.Ns)
        D = len(self.Ns)
        rdt = np.dtype(self.coord_dtype or self.dtype)     # the plan parameters are numbers of type real(Z)
        # sigma is converted to real(Z) first (src/plan.jl:573-576)
        sigma_wanted = float(rdt.type(self.sigma))
        self.Nover = tuple(
            oversampled_size(N, sigma_wanted, self.is_real and d == 0)
            for d, N in enumerate(self.Ns))
        for Nt in self.Nover:
            if Nt < 2 * self.M:   # check_nufft_size, src/plan.jl:545-556 (ArgumentError)
                raise ValueError(
                    f"data size is too small: sigma*N = {Nt} < {2 * self.M} = 2M")
        self.ks = init_wavenumbers(self.Ns, self.is_real)
        self.betas, self.taus, self.coefs, self.phihat, self.index_map = [], [], [], [], []
        for d in range(D):
            sigma_d = float(rdt.type(self.Nover[d] / self.Ns[d]))      # src/plan.jl:503
            k = self.ks[d]
            kk = np.fft.fftshift(k) if self.fftshift else k             # src/plan.jl:509-514
            tau = 0.0
            if self.kernel == KERNEL_BKB:
                beta = float(rdt.type(bkb_beta(self.M, sigma_d) if self.kernel_param is None else self.kernel_param))
                cs = bkb_poly_coefficients(self.M, beta)
                ph = bkb_fourier(kk, self.M, self.Nover[d], beta)
            elif self.kernel == KERNEL_KB:
                beta = float(rdt.type(kb_beta(self.M, sigma_d) if self.kernel_param is None else self.kernel_param))
                cs = kb_poly_coefficients(self.M, beta)
                ph = kb_fourier(kk, self.M, self.Nover[d], beta)
            elif self.kernel == KERNEL_GAUSSIAN:
                beta = float(rdt.type(gaussian_ell(self.M, sigma_d) if self.kernel_param is None else self.kernel_param))
                tau = float(rdt.type(gaussian_tau(beta, self.Nover[d])))
                cs = np.zeros((0, 2 * self.M))
                ph = gaussian_fourier(kk, tau)
            elif self.kernel == KERNEL_BSPLINE:
                beta = 0.0
                cs = np.zeros((0, 2 * self.M))
                ph = bspline_fourier(kk, self.M, self.Nover[d])
            else:
                raise ValueError(f"unknown kernel {self.kernel}")
            self.betas.append(beta)
            self.taus.append(tau)
            self.coefs.append(cs)
            self.phihat.append(ph)
            n_axis = self.Nover[d] // 2 + 1 if (self.is_real and d == 0) else self.Nover[d]
            self.index_map.append(non_oversampled_indices(k, n_axis, self.fftshift))

    # sizes -------------------------------------------------------------------------
    @property
    def ndim(self):
        return len(self.Ns)

    @property
    def size(self):
        """size(p): dims of the uniform (Fourier) arrays, src/plan.jl:426."""
        return tuple(len(k) for k in self.ks)

    @property
    def cdtype(self):
        return np.complex128 if np.dtype(self.dtype) == np.float64 else np.complex64

    @property
    def vdtype(self):
        """Element type Z of the non-uniform values."""
        return np.dtype(self.dtype) if self.is_real else np.dtype(self.cdtype)


# --------------------------------------------------------------------------------------
# Point folding, cell index, window evaluation
# --------------------------------------------------------------------------------------

def to_unit_cell(x: np.ndarray) -> np.ndarray:
    """Fold onto [0, 2pi) in the precision of ``x`` — GPU form, src/blocking/blocking.jl:26-33
    (``rem`` == C ``fmod``); equals the CPU while-loop form :12-21 wherever both are defined."""
    x = np.asarray(x)
    L = x.dtype.type(TWO_PI)
    r = np.fmod(x, L)
    r = np.where(r == 0, np.abs(r), r)          # -0.0 -> +0.0
    return np.where(r < 0, L + r, r).astype(x.dtype)


def nfft_point_convention(x: np.ndarray) -> np.ndarray:
    """AbstractNFFTs locations x in [-1/2, 1/2) and the opposite sign of the exponent -> x in [0, 2pi)
    (src/abstractNFFTs.jl:147-155), in the precision of ``x``."""
    x = np.asarray(x)
    L = x.dtype.type(TWO_PI)
    t = -(L * x)
    return np.where(t < 0, t + L, t).astype(x.dtype)


def point_to_cell(x: np.ndarray, N: int):
    """0-based cell index and r = (x / 2pi) * N in the precision of ``x``
    (src/Kernels/Kernels.jl:121-126; the order of operations matters, test/near_2pi.jl:37-45)."""
    x = np.asarray(x)
    T = x.dtype.type
    r = (x / T(TWO_PI)) * T(N)
    i = np.trunc(r).astype(np.int64)
    return i, r


def evaluate_window(plan: OraclePlan, d: int, x: np.ndarray):
    """Cell index (0-based) and the 2M window values of every point along dimension d.
    Values j = 0..2M-1 belong to grid nodes i - M + 1 + j (0-based), i.e. the reference's
    1-based ``(i - M + 1):(i + M)`` (src/Kernels/Kernels.jl:162-164).
    BackwardsKaiserBessel Direct: src/Kernels/kaiser_bessel_backwards.jl:158-175;
    FastApproximation (both Kaiser-Bessel kernels): :147-156 + src/Kernels/piecewise_polynomial.jl:76-92.
    The other kernels are dispatched below with their own citations."""
    M = plan.M
    T = x.dtype.type
    i, r = point_to_cell(x, plan.Nover[d])
    # r == N can only come from rounding of L + r in the fold (x = -tiny); the reference would index
    # out of bounds there.  Keep the point in the last cell with X = 1 (same window by continuity).
    i = np.minimum(i, plan.Nover[d] - 1)
    X = (r - i.astype(x.dtype)).astype(x.dtype)          # in [0, 1]
    if np.dtype(plan.dtype) != x.dtype:                  # coord_dtype: located in Float32, evaluated in Float64
        x = x.astype(plan.dtype)
        X = X.astype(plan.dtype)
        T = x.dtype.type
    if plan.kernel == KERNEL_BSPLINE:
        # src/Kernels/bspline.jl:99-119: x' = i - r (1-based i) = 1 - X; same recursion in both modes
        return i, bspline_evaluate_all((T(1) - X).astype(x.dtype), 2 * M).astype(x.dtype)
    if plan.kernel == KERNEL_GAUSSIAN:
        dx = T(TWO_PI) / T(plan.Nover[d])
        tau = T(plan.taus[d])
        if plan.evalmode == DIRECT:                       # src/Kernels/gaussian.jl:141-153
            js = np.arange(1, 2 * M + 1, dtype=x.dtype)
            ys = (T(M) - js[None, :] + X[:, None]) * dx
            return i, np.exp(-(ys * ys) / tau).astype(x.dtype)
        # fast Gaussian gridding, src/Kernels/gaussian.jl:125-139,155-192: a = exp(-X^2 / tau),
        # b = exp(2 X dx / tau) with X = x - (i - 1) dx in physical units, cs[m] = exp(-(m dx)^2 / tau)
        Xp = (x - i.astype(x.dtype) * dx).astype(x.dtype)
        a = np.exp(-(Xp * Xp) / tau).astype(x.dtype)
        b = np.exp(T(2) * Xp * dx / tau).astype(x.dtype)
        m = np.arange(1, M + 1, dtype=x.dtype)
        cs = np.exp(-((m * dx) ** 2) / tau).astype(x.dtype)
        vals = np.empty((len(x), 2 * M), dtype=x.dtype)
        vals[:, M - 1] = a
        bpow = np.ones_like(b)
        for mm in range(1, M):
            bpow = bpow * b
            vals[:, M - 1 - mm] = a * cs[mm - 1] / bpow
            vals[:, M - 1 + mm] = a * cs[mm - 1] * bpow
        vals[:, 2 * M - 1] = a * cs[M - 1] * bpow * b
        return i, vals
    if plan.kernel == KERNEL_KB and plan.evalmode == DIRECT:    # src/Kernels/kaiser_bessel.jl:197-210
        js = np.arange(1, 2 * M + 1, dtype=x.dtype)
        ys = (T(M) - js[None, :] + X[:, None]) / T(M)
        zs = np.maximum(T(1) - ys * ys, T(0))
        return i, _bessel_i0((T(plan.betas[d]) * np.sqrt(zs)).astype(np.float64)).astype(x.dtype)
    if plan.evalmode == DIRECT:
        js = np.arange(1, 2 * M + 1, dtype=x.dtype)
        ys = (T(M) - js[None, :] + X[:, None]) / T(M)
        zs = T(1) - ys * ys
        with np.errstate(invalid="ignore"):
            s = np.sqrt(zs)
        beta = T(plan.betas[d])
        bs = beta * s
        with np.errstate(divide="ignore", invalid="ignore"):
            vals = np.where(s == 0, T(1), np.sinh(bs) / bs) * (beta / T(np.pi))
        vals = vals.astype(x.dtype)
    else:
        cs = plan.coefs[d].astype(x.dtype)               # (npoly, 2M)
        xx = (T(2) * X - T(1))[:, None]
        vals = np.broadcast_to(cs[-1][None, :], (len(x), 2 * M)).astype(x.dtype)
        for k in range(cs.shape[0] - 2, -1, -1):         # Horner, piecewise_polynomial.jl:84-92
            vals = xx * vals + cs[k][None, :]
    return i, vals


# --------------------------------------------------------------------------------------
# The four stages
# --------------------------------------------------------------------------------------

def set_points(plan: OraclePlan, xp: Sequence[np.ndarray]):
    """set_points!: the oracle keeps the (unfolded) coordinates; folding happens per access
    exactly as ``point_transform_fold`` does (src/set_points.jl:33-52, src/plan.jl:459-464)."""
    if len(xp) != plan.ndim:
        raise ValueError(f"expected {plan.ndim}-dimensional points")
    n0 = len(xp[0])
    for x in xp:
        if np.asarray(x).dtype != np.dtype(plan.coord_dtype or plan.dtype):
            raise ValueError("input points must have the same accuracy as the created plan")
        if len(x) != n0:
            raise ValueError("input points must have the same length along all dimensions")
    plan.points = [np.ascontiguousarray(x) for x in xp]
    return plan


def _stencil_indices(plan: OraclePlan):
    """Per dimension: wrapped grid indices (Np, 2M) and window values (Np, 2M)."""
    inds, vals = [], []
    M = plan.M
    for d in range(plan.ndim):
        x = plan.points[d]
        if plan.point_transform == POINT_TRANSFORM_NFFT:      # to_unit_cell ∘ point_transform, src/set_points.jl:46-50
            x = nfft_point_convention(x)
        x = to_unit_cell(x)
        i, v = evaluate_window(plan, d, x)
        i = np.minimum(i, plan.Nover[d] - 1)
        j = (i[:, None] - M + 1 + np.arange(2 * M)[None, :]) % plan.Nover[d]   # kernel_indices
        inds.append(j)
        vals.append(v)
    return inds, vals


def spread(plan: OraclePlan, vp: Sequence[np.ndarray]):
    """Type-1 spreading u[wrap(i - M + j)] += v prod_d phi_d[j_d]
    (src/spreading/cpu_nonblocked.jl:16-93).  Returns C grids with reversed axes."""
    D = plan.ndim
    inds, vals = _stencil_indices(plan)
    shape = tuple(reversed(plan.Nover))
    gdtype = plan.vdtype
    L = 2 * plan.M
    # flattened (column-major in the reference == C-order with reversed axes) index
    lin = inds[0]                                   # (Np, L) dim 1 fastest
    w = vals[0]
    stride = plan.Nover[0]
    for d in range(1, D):
        lin = lin[:, None, ...] + stride * inds[d].reshape((-1, L) + (1,) * d)
        w = w[:, None, ...] * vals[d].reshape((-1, L) + (1,) * d)
        stride *= plan.Nover[d]
    Np = lin.shape[0]
    lin = lin.reshape(Np, -1)
    w = w.reshape(Np, -1)
    us = []
    for c in range(len(vp)):
        v = np.asarray(vp[c]).astype(gdtype)
        u = np.zeros(int(np.prod(shape)), dtype=gdtype)
        contrib = (v[:, None] * w).astype(gdtype)
        if np.iscomplexobj(u):
            ur = np.zeros(u.shape, dtype=plan.dtype)
            ui = np.zeros(u.shape, dtype=plan.dtype)
            np.add.at(ur, lin.ravel(), contrib.real.ravel())
            np.add.at(ui, lin.ravel(), contrib.imag.ravel())
            u = (ur + 1j * ui).astype(gdtype)
        else:
            np.add.at(u, lin.ravel(), contrib.ravel())
        us.append(u.reshape(shape))
    return us


def interpolate(plan: OraclePlan, us: Sequence[np.ndarray]):
    """Type-2 gather v = sum u[wrap(...)] prod_d (phi_d[j_d] dx_d)
    (src/interpolation/cpu_nonblocked.jl:26-79, src/interpolation/gpu.jl:55-56)."""
    D = plan.ndim
    inds, vals = _stencil_indices(plan)
    L = 2 * plan.M
    T = np.dtype(plan.dtype).type
    lin = inds[0]
    w = vals[0] * T(TWO_PI / plan.Nover[0])
    stride = plan.Nover[0]
    for d in range(1, D):
        lin = lin[:, None, ...] + stride * inds[d].reshape((-1, L) + (1,) * d)
        w = w[:, None, ...] * (vals[d] * T(TWO_PI / plan.Nover[d])).reshape((-1, L) + (1,) * d)
        stride *= plan.Nover[d]
    Np = lin.shape[0]
    lin = lin.reshape(Np, -1)
    w = w.reshape(Np, -1)
    out = []
    for u in us:
        uf = np.asarray(u).reshape(-1)
        out.append((uf[lin] * w).sum(axis=1).astype(plan.vdtype))
    return out


def _deconv_factor(plan: OraclePlan) -> np.ndarray:
    """prod_d phi_hat_d[I_d] on the output grid, reversed axes."""
    D = plan.ndim
    f = plan.phihat[0]
    for d in range(1, D):
        f = plan.phihat[d].reshape((-1,) + (1,) * d) * f
    return f


def _gather_index(plan: OraclePlan):
    return np.ix_(*[plan.index_map[d] for d in reversed(range(plan.ndim))])


class NUFFTCallbacks:
    """NUFFTCallbacks (src/plan.jl:146-164): `nonuniform(vs, n)` receives the tuple of the C values of point n
    (0-based here) and returns the tuple that is spread (type 1, src/spreading/cpu_nonblocked.jl:57-62) or stored
    (type 2, src/interpolation/cpu_nonblocked.jl:16-22); `uniform(ws, idx)` receives the tuple of the C deconvolved
    (type 1: and normalised) coefficients of output index idx (0-based, dimension 1 first) and returns the tuple that is
    written to the output (type 1, src/NonuniformFFTs.jl:372-379,394-401) or to the oversampled spectrum (type 2,
    :437-445,460-467).  Defaults return their first argument (`default_callback`, src/plan.jl:164)."""

    def __init__(self, nonuniform=None, uniform=None):
        self.nonuniform = nonuniform
        self.uniform = uniform


def _apply_nonuniform(cb, vps, dtype):
    """vs_new = callback(vs, i) for every point; results keep the element type (`oftype`, src/plan.jl:99-103)."""
    if cb is None or cb.nonuniform is None:
        return [np.asarray(v) for v in vps]
    Np = len(vps[0])
    out = [np.empty(Np, dtype=dtype) for _ in vps]
    for i in range(Np):
        new = cb.nonuniform(tuple(v[i] for v in vps), i)
        for c in range(len(vps)):
            out[c][i] = new[c]
    return out


def _apply_uniform(cb, ws, dtype):
    """w_new = callback(ws, idx) for every output index; arrays have reversed axes (dimension 1 last)."""
    if cb is None or cb.uniform is None:
        return ws
    out = [np.empty(w.shape, dtype=dtype) for w in ws]
    for I in np.ndindex(ws[0].shape):
        new = cb.uniform(tuple(w[I] for w in ws), tuple(reversed(I)))
        for c in range(len(ws)):
            out[c][I] = new[c]
    return out


def exec_type1(plan: OraclePlan, vp, return_grid: bool = False, callbacks=None):
    """exec_type1!: zero -> spread -> unnormalised forward FFT -> truncate + deconvolve + normalise
    (src/NonuniformFFTs.jl:148-189,197-211,350-385)."""
    single = not isinstance(vp, (list, tuple))
    vps = [vp] if single else list(vp)
    if len(vps) != plan.ntransforms:
        raise ValueError(f"wrong amount of data vectors (expected {plan.ntransforms})")
    Np = len(plan.points[0])
    for v in vps:
        if len(v) != Np:
            raise ValueError("wrong length of data vector")
    vps = _apply_nonuniform(callbacks, [np.asarray(v).astype(plan.vdtype) for v in vps], plan.vdtype)
    us = spread(plan, vps)
    norm = float(np.prod([TWO_PI / n for n in plan.Nover]))
    fac = norm / _deconv_factor(plan)
    outs = []
    for u in us:
        if plan.is_real:
            uh = np.fft.rfftn(u.astype(np.float64))          # r2c along the fastest axis == dim 1
        else:
            uh = np.fft.fftn(u.astype(np.complex128))
        w = uh[_gather_index(plan)] * fac
        outs.append(w.astype(plan.cdtype))
    outs = _apply_uniform(callbacks, outs, plan.cdtype)
    if return_grid:
        return (outs[0] if single else outs), us
    return outs[0] if single else outs


def exec_type2(plan: OraclePlan, uhat, return_grid: bool = False, callbacks=None):
    """exec_type2!: zero-pad + deconvolve -> unnormalised backward FFT -> interpolate
    (src/NonuniformFFTs.jl:237-314,416-451)."""
    single = not isinstance(uhat, (list, tuple))
    uhs = [uhat] if single else list(uhat)
    if len(uhs) != plan.ntransforms:
        raise ValueError(f"wrong amount of arrays (expected {plan.ntransforms})")
    expected = tuple(reversed(plan.size))
    fac = 1.0 / _deconv_factor(plan)
    grids = []
    for w in uhs:
        if np.asarray(w).shape != expected:
            raise ValueError(f"wrong dimensions of array (expected {plan.size})")
    # deconvolve, then the uniform callback, then the scatter into the zero-padded spectrum (src/NonuniformFFTs.jl:437-447)
    ws = [np.asarray(w).astype(np.complex128) * fac for w in uhs]
    if callbacks is not None and callbacks.uniform is not None:
        ws = _apply_uniform(callbacks, [w.astype(plan.cdtype) for w in ws], plan.cdtype)   # the callback sees values of type Z
    for w in ws:
        if plan.is_real:
            shape = tuple(reversed((plan.Nover[0] // 2 + 1,) + plan.Nover[1:]))
        else:
            shape = tuple(reversed(plan.Nover))
        uh = np.zeros(shape, dtype=np.complex128)
        uh[_gather_index(plan)] = w.astype(np.complex128)
        ntot = int(np.prod(plan.Nover))
        if plan.is_real:
            # brfft: unnormalised c2r of length Nover[0] along dim 1
            u = np.fft.irfftn(uh, s=tuple(reversed(plan.Nover)), axes=tuple(range(plan.ndim))) * ntot
            u = u.astype(plan.dtype)
        else:
            u = (np.fft.ifftn(uh) * ntot).astype(plan.cdtype)
        grids.append(u)
    vs = interpolate(plan, grids)
    vs = _apply_nonuniform(callbacks, vs, plan.vdtype)
    if return_grid:
        return (vs[0] if single else vs), grids
    return vs[0] if single else vs


# --------------------------------------------------------------------------------------
# Exact transforms (the reference tests' known answers)
# --------------------------------------------------------------------------------------

def nudft_type1(ks_list, xp, vp) -> np.ndarray:
    """u_hat(k) = sum_j v_j exp(-i k.x_j) (docs/src/index.md:28-58; test/accuracy.jl:119-125).
    Output has reversed axes (dimension 1 fastest)."""
    D = len(ks_list)
    xs = [np.asarray(x, dtype=np.float64) for x in xp]
    v = np.asarray(vp).astype(np.complex128)
    E = [np.exp(-1j * np.outer(ks_list[d], xs[d])) for d in range(D)]     # (Nk_d, Np)
    if D == 1:
        return E[0] @ v
    if D == 2:
        return np.einsum("bp,ap,p->ba", E[1], E[0], v, optimize=True)
    return np.einsum("cp,bp,ap,p->cba", E[2], E[1], E[0], v, optimize=True)


def nudft_type2(ks_list, xp, uhat) -> np.ndarray:
    """v_j = sum_k u_hat(k) exp(+i k.x_j) (docs/src/index.md:28-58; test/accuracy.jl:180-188)."""
    D = len(ks_list)
    xs = [np.asarray(x, dtype=np.float64) for x in xp]
    E = [np.exp(1j * np.outer(ks_list[d], xs[d])) for d in range(D)]
    uh = np.asarray(uhat).astype(np.complex128)
    if D == 1:
        return uh @ E[0]
    if D == 2:
        return np.einsum("ba,bp,ap->p", uh, E[1], E[0], optimize=True)
    return np.einsum("cba,cp,bp,ap->p", uh, E[2], E[1], E[0], optimize=True)


def hermitian_weights(plan: OraclePlan) -> np.ndarray:
    """For real-data plans the r2c output keeps only k1 >= 0; a real type-2 transform implicitly
    uses u_hat(-k) = conj(u_hat(k)).  Weight 1 for k1 = 0 and 2 otherwise — including k1 = N/2,
    which is an ordinary (non-Nyquist) mode of the *oversampled* c2r transform
    (test/accuracy.jl:184-186: ``factor = ifelse(iszero(k), 1, 2)``)."""
    w = np.full(len(plan.ks[0]), 2.0)
    w[0] = 1.0
    return w


def nudft_type2_real(plan: OraclePlan, xp, uhat) -> np.ndarray:
    """Exact type-2 for a real-data plan: v_j = Re sum_k w(k1) u_hat(k) exp(i k.x_j), which is what
    brfft-based evaluation computes (test/accuracy.jl:180-196 does the same with explicit
    conjugate pairs)."""
    w = hermitian_weights(plan)
    uh = np.asarray(uhat).astype(np.complex128) * w      # broadcast over the fastest axis
    return nudft_type2(plan.ks, xp, uh).real


def l2_error(a, b) -> float:
    """test/accuracy.jl:76-82."""
    a = np.asarray(a)
    b = np.asarray(b)
    return float(np.sqrt(np.sum(np.abs(a - b) ** 2) / np.sum(np.abs(b) ** 2)))
