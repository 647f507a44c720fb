"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on identical seeded
inputs.  Mirrors the reference's GPU-vs-CPU matrix (test/pseudo_gpu.jl:109-226): dims (35, 64, 40),
Np = prod(dims) there; smaller Np here so the numpy oracle finishes in seconds.

Tolerances (written here as the reference writes them): rtol 1e-7 for Float64 and 1e-5 for Float32 on
the 2-norm (test/pseudo_gpu.jl:159-171).  Summation order inside a tile is nondeterministic on the
GPU (atomics), so comparisons are never bitwise.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import nufft_oracle as O  # noqa: E402


def _nufft():
    from nufft_pkg import nufft
    return nufft


def _rtol(dtype):
    return 1e-7 if np.dtype(dtype) in (np.dtype(np.float64), np.dtype(np.complex128)) else 1e-5


def plan_real_dtype(Z):
    return np.float32 if np.dtype(Z) in (np.dtype(np.float32), np.dtype(np.complex64)) else np.float64


def _rel(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    return float(np.linalg.norm((a - b).ravel()) / np.linalg.norm(b.ravel()))


def _f32_overflows(Z, dims, M):
    """The reference's un-normalised BKB window peaks at e^β/2π; in Float32 the product of D window
    values overflows for large M (D = 3: M >= 7 at sigma = 2; D = 2: M = 10 — β = 46.9 — and, with the sums of a
    cell, M = 9: D * M >= 18 is the conservative rule).  There the Float32 oracle is not finite and the HIP
    path (which normalises the window by an exact power of two) is checked against the Float64 oracle that locates
    the points in Float32 exactly as a Float32 plan does (`coord_dtype`), at the reference's Float32 bound 1e-5."""
    return np.dtype(Z) in (np.dtype(np.float32), np.dtype(np.complex64)) and len(dims) * M >= 18


_KERNEL_OBJ = {O.KERNEL_BKB: "BackwardsKaiserBesselKernel", O.KERNEL_KB: "KaiserBesselKernel",
               O.KERNEL_GAUSSIAN: "GaussianKernel", O.KERNEL_BSPLINE: "BSplineKernel"}


def _make_case(Z, dims, M, sigma, evalmode, C, Np, seed, kernel=O.KERNEL_BKB, kernel_param=None, fftshift=False,
               point_transform=0, **kw):
    nufft = _nufft()
    Z = np.dtype(Z)
    is_real = Z.kind == "f"
    T = np.dtype(np.float32) if Z.itemsize in (4,) or Z == np.complex64 else np.dtype(np.float64)
    rng = np.random.default_rng(seed)
    xs = [(rng.random(Np) * 3 - 1) * O.TWO_PI for _ in dims]          # points outside the unit cell too
    if point_transform == O.POINT_TRANSFORM_NFFT:
        xs = [rng.random(Np) - 0.5 for _ in dims]                     # AbstractNFFTs convention: [-1/2, 1/2)
    xs = [x.astype(T) for x in xs]
    if is_real:
        vs = [rng.standard_normal(Np).astype(Z) for _ in range(C)]
    else:
        vs = [(rng.standard_normal(Np) + 1j * rng.standard_normal(Np)).astype(Z) for _ in range(C)]
    mode = nufft.Direct() if evalmode == O.DIRECT else nufft.FastApproximation()
    kcls = getattr(nufft, _KERNEL_OBJ[kernel])
    kobj = kcls() if kernel_param is None else kcls(kernel_param)
    plan = nufft.PlanNUFFT(Z, dims, m=M, sigma=sigma, ntransforms=C, kernel_evalmode=mode, kernel=kobj,
                           fftshift=fftshift, point_transform="nfft" if point_transform else None,
                           backend=nufft.ROCBackend(0), **kw)
    big_window = kernel in (O.KERNEL_BKB, O.KERNEL_KB)
    wide = big_window and _f32_overflows(Z, dims, M)
    oplan = O.OraclePlan(dims, is_real=is_real, dtype=np.float64 if wide else T.type, coord_dtype=T.type if wide else None,
                         M=M, sigma=sigma, evalmode=evalmode, ntransforms=C,
                         kernel=kernel, kernel_param=kernel_param, fftshift=fftshift, point_transform=point_transform)
    return nufft, plan, oplan, xs, vs


def _oracle_inputs(oplan, arrs):
    """Values / spectra in the oracle's precision (identity unless the Float64 oracle stands in for Float32; the
    coordinates always keep the plan's precision, see `coord_dtype`)."""
    if np.dtype(oplan.dtype) == np.float64:
        return [a.astype(np.complex128 if np.iscomplexobj(a) else np.float64) for a in arrs]
    return arrs


CASES = [
    # Z, dims, M, sigma, evalmode, C
    (np.float64, (35, 64, 40), 4, 1.5, O.DIRECT, 1),
    (np.float64, (35, 64, 40), 4, 1.5, O.FAST_APPROXIMATION, 1),
    (np.complex128, (35, 64, 40), 4, 1.5, O.DIRECT, 1),
    (np.float32, (35, 64, 40), 4, 1.5, O.DIRECT, 1),
    (np.complex64, (35, 64, 40), 4, 1.5, O.FAST_APPROXIMATION, 1),
    (np.float64, (35, 64, 40), 4, 2.0, O.DIRECT, 2),              # ntransforms = 2
    (np.complex128, (24, 20, 30), 6, 2.0, O.FAST_APPROXIMATION, 3),
    (np.float64, (64, 64), 4, 2.0, O.DIRECT, 1),
    (np.complex128, (37, 41), 5, 1.25, O.FAST_APPROXIMATION, 1),
    (np.float32, (64, 48), 2, 2.0, O.DIRECT, 2),
    (np.float64, (256,), 4, 2.0, O.DIRECT, 1),                    # BASELINE config C1 shape
    (np.complex128, (256,), 8, 1.25, O.FAST_APPROXIMATION, 1),
    (np.complex64, (100,), 3, 2.0, O.DIRECT, 1),
    (np.float64, (20, 16, 18), 8, 2.0, O.DIRECT, 1),              # wide support
    (np.complex64, (16, 16, 16), 8, 2.0, O.FAST_APPROXIMATION, 1),
    (np.float64, (12, 16, 10), 10, 2.0, O.FAST_APPROXIMATION, 1),
    (np.float64, (30, 30, 30), 3, 2.0, O.DIRECT, 1),              # odd half-support
    # power-of-two oversampled grids: the pruned FFT passes (fft_lines.hip) replace rocFFT's dims 2, 3
    (np.float64, (32, 32, 32), 4, 2.0, O.DIRECT, 1),
    (np.float64, (31, 31, 31), 4, 2.0, O.FAST_APPROXIMATION, 2),  # odd N_out, two transforms
    (np.float32, (32, 64, 32), 4, 2.0, O.FAST_APPROXIMATION, 1),
    (np.float64, (64, 128), 6, 2.0, O.DIRECT, 1),                 # 2-D: 128 x 256
    (np.float64, (48, 128, 32), 4, 2.0, O.DIRECT, 1),             # dim 1 not a power of two (96), dims 2, 3 are
    # oversampled sizes 1.5 * 2^a and 1.25 * 2^a: radix-3 / radix-5 stages of the pruned passes
    (np.float64, (64, 64, 64), 4, 1.5, O.FAST_APPROXIMATION, 1),  # 96^3
    (np.float64, (64, 128, 64), 4, 1.25, O.DIRECT, 2),            # 80 x 160 x 80, two transforms
    (np.float32, (128, 256), 4, 1.5, O.FAST_APPROXIMATION, 1),    # 192 x 384
    (np.float64, (256, 64, 32), 6, 1.5, O.DIRECT, 1),             # 384 x 96 x 48 (48: general path for that plan)
    (np.float64, (63, 64, 128), 5, 1.5, O.FAST_APPROXIMATION, 1), # odd N1: 2 * nextprod(48) = 96, N_out1 = 32
    # complex plans on the pruned path (own c2c pass along dimension 1 with a compact spectrum)
    (np.complex128, (32, 64, 32), 4, 2.0, O.DIRECT, 1),
    (np.complex64, (64, 64), 4, 1.5, O.FAST_APPROXIMATION, 1),    # 96 x 96
    (np.complex128, (64, 32, 64), 6, 1.25, O.FAST_APPROXIMATION, 2),  # 80 x 40(general: 40 unsupported -> whole plan general)
    (np.complex128, (64, 64, 64), 4, 1.25, O.DIRECT, 2),          # 80^3, two transforms
    (np.complex64, (33, 64, 47), 4, 2.0, O.DIRECT, 1),            # odd sizes: 66 (general)
    (np.complex128, (127, 128), 5, 2.0, O.FAST_APPROXIMATION, 1), # odd N1 = 127 -> 254 (general); 
    (np.complex128, (128, 127), 5, 2.0, O.FAST_APPROXIMATION, 1),
    (np.complex128, (48, 96), 5, 2.0, O.FAST_APPROXIMATION, 1),   # 96 x 192
    # wide support on multi-tile 3-D grids: every tile sees n / b + ceil(M / b) + floor((M - 2) / b) + 1 bin rows per
    # dimension (the work-item table was sized for n / b + 4 and silently dropped runs for M >= 9)
    (np.float64, (64, 64, 64), 10, 2.0, O.FAST_APPROXIMATION, 1),
    (np.float64, (64, 64, 64), 9, 2.0, O.DIRECT, 1),
    (np.float32, (64, 64, 64), 10, 2.0, O.FAST_APPROXIMATION, 1),
    (np.complex64, (64, 64, 64), 10, 2.0, O.DIRECT, 1),
    (np.complex128, (64, 64, 64), 10, 2.0, O.FAST_APPROXIMATION, 1),   # 2-cell bins: n / 2 + 10 bin rows
    (np.complex128, (48, 40, 56), 9, 1.5, O.DIRECT, 1),
    # Float64 lines of 2 x 1024 in the dimension-1 r2c / c2r pass (4 lines per workgroup: 8 need 172 KB of LDS)
    (np.float64, (1024, 32), 4, 2.0, O.DIRECT, 1),
    (np.float32, (1024, 32), 4, 2.0, O.FAST_APPROXIMATION, 1),
]


# The other spreading kernels: every kernel x evaluation mode of the reference
# (test/accuracy.jl:252-283 loops over the same four kernels), all element types, D = 1..3.
KERNEL_CASES = [
    # kernel, Z, dims, M, sigma, evalmode, C
    (O.KERNEL_KB, np.float64, (35, 64, 40), 4, 1.5, O.DIRECT, 1),
    (O.KERNEL_KB, np.float64, (35, 64, 40), 4, 1.5, O.FAST_APPROXIMATION, 1),
    (O.KERNEL_KB, np.complex64, (24, 20, 30), 4, 2.0, O.DIRECT, 1),
    (O.KERNEL_KB, np.complex128, (37, 41), 8, 1.25, O.DIRECT, 2),
    (O.KERNEL_KB, np.float32, (100,), 2, 2.0, O.FAST_APPROXIMATION, 1),
    (O.KERNEL_KB, np.float64, (16, 12, 14), 10, 2.0, O.DIRECT, 1),
    (O.KERNEL_GAUSSIAN, np.float64, (35, 64, 40), 4, 2.0, O.DIRECT, 1),
    (O.KERNEL_GAUSSIAN, np.float64, (35, 64, 40), 4, 2.0, O.FAST_APPROXIMATION, 1),
    (O.KERNEL_GAUSSIAN, np.complex128, (24, 20, 30), 7, 2.0, O.FAST_APPROXIMATION, 2),
    (O.KERNEL_GAUSSIAN, np.float32, (64, 48), 3, 2.0, O.DIRECT, 1),
    (O.KERNEL_GAUSSIAN, np.complex64, (128,), 5, 1.5, O.FAST_APPROXIMATION, 1),
    (O.KERNEL_GAUSSIAN, np.float64, (16, 12, 14), 10, 2.0, O.FAST_APPROXIMATION, 1),
    (O.KERNEL_BSPLINE, np.float64, (35, 64, 40), 4, 2.0, O.DIRECT, 1),
    (O.KERNEL_BSPLINE, np.float64, (32, 32, 32), 4, 2.0, O.FAST_APPROXIMATION, 1),
    (O.KERNEL_BSPLINE, np.complex128, (24, 20, 30), 6, 2.0, O.DIRECT, 2),
    (O.KERNEL_BSPLINE, np.float32, (64, 48), 2, 2.0, O.FAST_APPROXIMATION, 1),
    (O.KERNEL_BSPLINE, np.complex64, (100,), 3, 1.25, O.DIRECT, 1),
    (O.KERNEL_BSPLINE, np.float64, (16, 12, 14), 10, 2.0, O.DIRECT, 1),
]


@pytest.mark.parametrize("kernel,Z,dims,M,sigma,evalmode,C", KERNEL_CASES)
def test_other_kernels_match_oracle(kernel, Z, dims, M, sigma, evalmode, C):
    _check_type1_type2(Z, dims, M, sigma, evalmode, C, kernel=kernel)


def test_explicit_kernel_parameters_match_oracle():
    """KaiserBesselKernel(β), BackwardsKaiserBesselKernel(β), GaussianKernel(ℓ) (test/accuracy.jl:285-297)."""
    beta = 2 * np.pi * (2 - 1 / 2.0)
    _check_type1_type2(np.float64, (48, 40), 2, 2.0, O.DIRECT, 1, kernel=O.KERNEL_KB, kernel_param=beta)
    _check_type1_type2(np.float64, (48, 40), 2, 2.0, O.FAST_APPROXIMATION, 1, kernel=O.KERNEL_BKB, kernel_param=beta)
    _check_type1_type2(np.complex128, (48, 40), 4, 2.0, O.FAST_APPROXIMATION, 1, kernel=O.KERNEL_GAUSSIAN, kernel_param=1.05)


@pytest.mark.parametrize("Z,dims", [(np.float64, (35, 64, 40)), (np.complex128, (37, 41)), (np.float64, (32, 32, 32)),
                                    (np.float64, (48, 48, 48)), (np.complex128, (32, 64, 48)), (np.complex64, (31, 32)), (np.complex128, (64, 96)),
                                    (np.complex64, (100,)), (np.float64, (31, 33)), (np.complex128, (16, 15, 12))])
def test_fftshift_ordering_matches_oracle(Z, dims):
    """fftshift = true: uniform data in increasing-frequency order (src/plan.jl:472,509-514,
    src/NonuniformFFTs.jl:318-348), on the general and on the pruned-FFT path, odd and even sizes."""
    _check_type1_type2(Z, dims, 4, 2.0, O.FAST_APPROXIMATION, 1, fftshift=True)


@pytest.mark.parametrize("Z,dims,fftshift", [(np.complex128, (64, 81), True), (np.complex64, (128,), True),
                                             (np.complex128, (24, 20, 30), False), (np.float64, (40, 36), True)])
def test_nfft_point_convention_matches_oracle(Z, dims, fftshift):
    """point_transform = _transform_point_convention (src/abstractNFFTs.jl:147-155) on the device."""
    _check_type1_type2(Z, dims, 4, 2.0, O.DIRECT, 1, fftshift=fftshift, point_transform=O.POINT_TRANSFORM_NFFT)


def test_nfft_plan_interface():
    """NonuniformFFTs.NFFTPlan / plan_nfft (src/abstractNFFTs.jl:52-229) against the direct sums of the
    NFFT definition (test/abstractNFFTs.jl:11-64 compares with NFFT.jl on the same sizes)."""
    nufft = _nufft()
    rng = np.random.default_rng(43)
    dims, Np = (64, 81), 1000
    xp = rng.random((Np, 2)) - 0.5
    vp = rng.standard_normal(Np) + 1j * rng.standard_normal(Np)
    xd = torch.from_numpy(xp).cuda()
    p = nufft.plan_nfft(xd, dims, m=5, sigma=2.0, window="kaiser_bessel")
    assert isinstance(p, nufft.NFFTPlan) and p.size_in == dims and p.size_out == (Np,)
    assert repr(p).startswith("NonuniformFFTs.NFFTPlan{torch.float64, 2} wrapping a PlanNUFFT:")
    us = p.adjoint_mul(torch.from_numpy(vp).cuda())
    ks = [np.arange(-(N // 2), -(N // 2) + N, dtype=np.float64) for N in dims]
    E = [np.exp(2j * np.pi * np.outer(ks[d], xp[:, d])) for d in range(2)]
    exact = np.einsum("bp,ap,p->ba", E[1], E[0], vp, optimize=True)
    assert _rel(us.cpu().numpy(), exact) < 1e-9
    wp = p.mul(torch.from_numpy(np.ascontiguousarray(exact)).cuda())
    exact2 = np.einsum("bp,ap,ba->p", np.conj(E[1]), np.conj(E[0]), exact, optimize=True)
    assert _rel(wp.cpu().numpy(), exact2) < 1e-9
    # the arguments the reference accepts for scheduling only are accepted here too
    q = nufft.PlanNUFFT(np.complex128, (24, 24), gpu_method="global_memory", sort_points=True, block_size=(8, 8))
    assert q.size == (24, 24)




@pytest.mark.parametrize("Z,Ns,C,M", [(np.float32, (64, 32, 16), 1, 4), (np.complex64, (64, 32, 16), 1, 4),
                                      (np.complex128, (32, 32, 32), 2, 4), (np.float64, (32, 32, 16), 2, 4),
                                      (np.complex128, (40, 24), 1, 6), (np.float64, (128,), 1, 4),
                                      (np.complex64, (32, 32, 32), 1, 8)])
def test_callbacks_match_oracle(Z, Ns, C, M):
    """SURVEY §8(f) row 3 against the ORACLE's model of NUFFTCallbacks (oracle/nufft_oracle.py, restating
    src/plan.jl:146-164 and the call sites src/spreading/cpu_nonblocked.jl:57-62, src/interpolation/cpu_nonblocked.jl:16-22,
    src/NonuniformFFTs.jl:372-379,437-447): the callbacks of test/callbacks.jl:17-25 (per-point weights; 1/k², 0 at k = 0)
    as Python functions inside the oracle, as the fused menu (nufft_exec_type{1,2}_cb) on the device.  Covers the
    reference's two cases (Float32 / ComplexF32, Ns = (64, 32, 16), Np = prod(Ns) ÷ 3), ntransforms = 2, 2-D, 1-D, the
    pruned-FFT and the general path, and a plan of the MFMA-patch engine (per-point weights take its LDS-tile fallback)."""
    nufft = _nufft()
    Zt = np.dtype(Z)
    is_real = Zt.kind == "f"
    T = plan_real_dtype(Z)
    Np = int(np.prod(Ns)) // 3
    rng = np.random.default_rng(42)
    weights = rng.random(Np).astype(T)
    ks = [(np.fft.rfftfreq(N, 1 / N) if (d == 0 and is_real) else np.fft.fftfreq(N, 1 / N)) for d, N in enumerate(Ns)]
    k2 = sum(np.reshape(k ** 2, [-1 if e == d else 1 for e in range(len(Ns))][::-1]) for d, k in enumerate(ks))
    factors = np.where(k2 == 0, 0.0, 1.0 / np.where(k2 == 0, 1.0, k2)).astype(T)     # reversed axes = torch layout
    xs = [(rng.random(Np) * 2 * np.pi).astype(T) for _ in Ns]
    vs = [(rng.standard_normal(Np) if is_real else rng.standard_normal(Np) + 1j * rng.standard_normal(Np)).astype(Zt)
          for _ in range(C)]
    plan = nufft.PlanNUFFT(Zt, Ns, m=M, ntransforms=C, backend=nufft.ROCBackend(0))
    wide = _f32_overflows(Z, Ns, M)
    oplan = O.OraclePlan(Ns, is_real=is_real, dtype=np.float64 if wide else T, coord_dtype=T if wide else None, M=M,
                         sigma=2.0, evalmode=O.DIRECT, ntransforms=C)                 # Direct(): the ROC default
    ocb = O.NUFFTCallbacks(nonuniform=lambda v, n: tuple(type(x)(x * weights[n]) for x in v),
                           uniform=lambda w, idx: tuple(type(x)(x * factors[tuple(reversed(idx))]) for x in w))
    dev = plan.device
    nufft.set_points(plan, tuple(torch.from_numpy(x).to(dev) for x in xs))
    O.set_points(oplan, xs)
    wd, fd = torch.from_numpy(weights).to(dev), torch.from_numpy(np.ascontiguousarray(factors)).to(dev)
    cb = nufft.NUFFTCallbacks(nonuniform=nufft.PointWeights(wd), uniform=nufft.ModeFactors(fd))
    tup = (lambda t: t if C > 1 else t[0])
    vd = tuple(torch.from_numpy(v).to(dev) for v in vs)
    ws = tuple(torch.empty(plan.shape, dtype=plan.eltype, device=dev) for _ in range(C))
    nufft.exec_type1(tup(ws), plan, tup(vd), callbacks=cb)
    ref1 = O.exec_type1(oplan, _oracle_inputs(oplan, vs), callbacks=ocb)
    tol = _rtol(Z)
    for c in range(C):
        assert _rel(ws[c].cpu().numpy(), ref1[c]) < tol
    wp = tuple(torch.empty(Np, dtype=plan.Z, device=dev) for _ in range(C))
    nufft.exec_type2(tup(wp), plan, tup(ws), callbacks=cb)
    ref2 = O.exec_type2(oplan, _oracle_inputs(oplan, [w.cpu().numpy() for w in ws]), callbacks=ocb)
    for c in range(C):
        assert _rel(wp[c].cpu().numpy(), ref2[c]) < tol
    # one callback at a time
    only_u = nufft.NUFFTCallbacks(uniform=nufft.ModeFactors(fd))
    w1 = tuple(torch.empty(plan.shape, dtype=plan.eltype, device=dev) for _ in range(C))
    nufft.exec_type1(tup(w1), plan, tup(vd), callbacks=only_u)
    r1 = O.exec_type1(oplan, _oracle_inputs(oplan, vs), callbacks=O.NUFFTCallbacks(uniform=ocb.uniform))
    assert _rel(w1[0].cpu().numpy(), r1[0]) < tol
    only_n = nufft.NUFFTCallbacks(nonuniform=nufft.PointWeights(wd))
    p1 = tuple(torch.empty(Np, dtype=plan.Z, device=dev) for _ in range(C))
    nufft.exec_type2(tup(p1), plan, tup(ws), callbacks=only_n)
    r2 = O.exec_type2(oplan, _oracle_inputs(oplan, [w.cpu().numpy() for w in ws]), callbacks=O.NUFFTCallbacks(nonuniform=ocb.nonuniform))
    assert _rel(p1[0].cpu().numpy(), r2[0]) < tol


@pytest.mark.parametrize("Z,dims,M,sigma,evalmode,C", CASES)
def test_type1_type2_match_oracle(Z, dims, M, sigma, evalmode, C):
    _check_type1_type2(Z, dims, M, sigma, evalmode, C)


# Both spreading engines (LDS tiles with ds_add_f64; register patches accumulated by v_mfma_f64_4x4x4) on the same
# 3-D cases, whatever the automatic choice would be: every half-support (the cube ring of the patches is 3..7 layers
# deep for M = 2..10), all element types, ntransforms > 1, both window evaluations.
ENGINE_CASES = [
    (np.float64, (32, 32, 32), 4, 2.0, O.DIRECT, 1),
    (np.float64, (64, 64, 64), 4, 1.5, O.FAST_APPROXIMATION, 2),     # 96^3
    (np.float32, (32, 64, 32), 4, 2.0, O.FAST_APPROXIMATION, 1),
    (np.complex128, (32, 64, 32), 4, 2.0, O.DIRECT, 1),
    (np.complex64, (64, 64, 64), 8, 2.0, O.FAST_APPROXIMATION, 1),   # BASELINE C3's (element type, M)
    (np.float64, (64, 64, 64), 10, 2.0, O.FAST_APPROXIMATION, 1),
    (np.float64, (48, 128, 32), 6, 2.0, O.DIRECT, 1),                # 96 x 256 x 64
    (np.float64, (40, 40, 40), 2, 2.0, O.DIRECT, 1),
    (np.float64, (40, 40, 40), 3, 2.0, O.FAST_APPROXIMATION, 1),
    (np.complex128, (40, 40, 40), 5, 2.0, O.FAST_APPROXIMATION, 1),
    (np.float32, (40, 40, 40), 7, 2.0, O.FAST_APPROXIMATION, 1),
    (np.float64, (40, 40, 40), 9, 2.0, O.DIRECT, 1),
    (np.complex64, (40, 48, 40), 6, 2.0, O.DIRECT, 3),
    # 72 x 100 x 80: 18 x 25 x 20 bins — the last patch column and the last patch row are partial
    (np.float64, (36, 50, 40), 4, 2.0, O.DIRECT, 1),
    (np.complex128, (36, 50, 40), 4, 2.0, O.FAST_APPROXIMATION, 1),
    (np.float32, (36, 50, 40), 5, 2.0, O.FAST_APPROXIMATION, 2),
    # real plans with ntransforms = 2 / 3: the patch engine spreads the components together (planar components)
    (np.float64, (32, 32, 32), 4, 2.0, O.DIRECT, 3),
    (np.float32, (40, 48, 40), 6, 2.0, O.FAST_APPROXIMATION, 3),
    (np.float64, (36, 50, 40), 4, 2.0, O.FAST_APPROXIMATION, 2),
    (np.float64, (40, 40, 40), 2, 2.0, O.DIRECT, 3),
    # wide supports in Float64 (the Float32 cases above are bounded by Float32 round-off): the same code at 1e-7
    (np.complex128, (48, 48, 48), 8, 2.0, O.FAST_APPROXIMATION, 1),
    (np.float64, (48, 48, 48), 8, 2.0, O.FAST_APPROXIMATION, 1),
    (np.complex128, (48, 48, 48), 8, 2.0, O.DIRECT, 1),
]


@pytest.mark.parametrize("engine", ["lds_tiles", "mfma_patches"])
@pytest.mark.parametrize("Z,dims,M,sigma,evalmode,C", ENGINE_CASES)
def test_both_spreading_engines_match_oracle(engine, Z, dims, M, sigma, evalmode, C):
    _check_type1_type2(Z, dims, M, sigma, evalmode, C, spread_method=engine, expect_engine=engine)


@pytest.mark.parametrize("M", range(2, 11))
@pytest.mark.parametrize("Z", [np.float32, np.complex64, "complex64/f64acc", np.float64, np.complex128])
def test_patch_engine_every_instantiation(Z, M, monkeypatch):
    """Every (element type, M) instantiation of the patch kernels against the oracle (type 1; ADVICE round 2: only 15 of
    the 36 were exercised).  96^3 oversampled grid = 24 bins per axis: partial last patch rows.  ComplexF32 has two
    kernels: Float32 accumulators on v_mfma_f32_16x16x4 (the default where dimension 3 is a multiple of 8 cells) and
    Float64 accumulators on v_mfma_f64_4x4x4 (NUFFT_PATCH_F32ACC=0 here; other grids in ENGINE_CASES)."""
    f32acc = None
    if isinstance(Z, str):
        Z, f32acc = np.complex64, 0
        monkeypatch.setenv("NUFFT_PATCH_F32ACC", "0")
    elif np.dtype(Z) == np.complex64:
        f32acc = 1
    if np.dtype(Z) == np.complex128 and M == 10:
        # the LDS-tile engines of this plan need 2-cell bins (tile + work-item table in 160 KiB); the patches serve 4-cell bins
        with pytest.raises(ValueError):
            _make_case(Z, (48, 48, 48), M, 2.0, O.FAST_APPROXIMATION, 1, 1500, seed=11 + M, spread_method="mfma_patches")
        return
    nufft, plan, oplan, xs, vs = _make_case(Z, (48, 48, 48), M, 2.0, O.FAST_APPROXIMATION, 1, 1500, seed=11 + M,
                                            spread_method="mfma_patches")
    assert plan.info().spread_method == 2
    if f32acc is not None:
        assert plan.info().patch_f32acc == f32acc
    dev = plan.device
    nufft.set_points(plan, tuple(torch.from_numpy(x).to(dev) for x in xs))
    O.set_points(oplan, xs)
    u = torch.empty(plan.shape, dtype=plan.eltype, device=dev)
    nufft.exec_type1(u, plan, torch.from_numpy(vs[0]).to(dev))
    assert plan.spread_engine_used() == "mfma_patches"
    ref = O.exec_type1(oplan, _oracle_inputs(oplan, vs)[0])
    assert _rel(u.cpu().numpy(), ref) < _rtol(Z)


@pytest.mark.parametrize("M", range(2, 11))
@pytest.mark.parametrize("Z", [np.float32, np.complex64, np.float64, np.complex128])
def test_lds_tile_spreading_every_instantiation(Z, M):
    """Every (element type, M) instantiation of spread_tile_kernel on a 3-D grid larger than its tile (compile-time tile
    variants where the plan has them) against the oracle, type 1, both window evaluations."""
    dims, Np = (48, 40, 56), 3000
    for evalmode in (O.FAST_APPROXIMATION, O.DIRECT):
        nufft, plan, oplan, xs, vs = _make_case(Z, dims, M, 2.0, evalmode, 1, Np, seed=300 + M, spread_method="lds_tiles")
        assert plan.info().spread_method == 1
        dev = plan.device
        nufft.set_points(plan, tuple(torch.from_numpy(x).to(dev) for x in xs))
        O.set_points(oplan, xs)
        u = torch.empty(plan.shape, dtype=plan.eltype, device=dev)
        nufft.exec_type1(u, plan, torch.from_numpy(vs[0]).to(dev))
        assert plan.spread_engine_used() == "lds_tiles"
        ref = O.exec_type1(oplan, _oracle_inputs(oplan, vs)[0])
        assert _rel(u.cpu().numpy(), ref) < _rtol(Z), evalmode


@pytest.mark.parametrize("M", range(2, 7))
@pytest.mark.parametrize("C", [2, 3])
@pytest.mark.parametrize("Z", [np.float32, np.float64])
def test_planar_patch_every_instantiation(Z, C, M):
    """Every (element type, ntransforms, M) instantiation of the patch kernel with planar components (one set-up for the
    C value vectors of a real plan) against the oracle, type 1; 96 x 80 x 112 oversampled: partial patch rows."""
    Np = 2500
    nufft, plan, oplan, xs, vs = _make_case(Z, (48, 40, 56), M, 2.0, O.FAST_APPROXIMATION, C, Np, seed=200 + 10 * C + M,
                                            spread_method="mfma_patches")
    assert plan.info().spread_method == 2 and plan.info().patch_planar == C
    dev = plan.device
    nufft.set_points(plan, tuple(torch.from_numpy(x).to(dev) for x in xs))
    O.set_points(oplan, xs)
    us = tuple(torch.empty(plan.shape, dtype=plan.eltype, device=dev) for _ in range(C))
    nufft.exec_type1(us, plan, tuple(torch.from_numpy(v).to(dev) for v in vs))
    assert plan.spread_engine_used() == "mfma_patches"
    ref = O.exec_type1(oplan, _oracle_inputs(oplan, vs))
    for c in range(C):
        assert _rel(us[c].cpu().numpy(), ref[c]) < _rtol(Z), c


@pytest.mark.parametrize("M", range(2, 11))
@pytest.mark.parametrize("Z", [np.float32, np.complex64, np.float64, np.complex128])
def test_interpolation_ring_every_instantiation(Z, M, monkeypatch):
    """Every (element type, M) instantiation of interp_march_kernel against the oracle (type 2, both window evaluations).
    Oversampled grid 96 x 80 x 112: partial columns at the upper ends of dimensions 1 and 2, several segments along
    dimension 3.  On a grid this small the ring's few tasks cannot fill the chip and set_points gives the point set to
    the LDS-tile kernel, so the test plans force the ring (NUFFT_INTERP_MARCH=2) and nufft_interp_engine_used confirms the
    device-side flag; a point set concentrated in a corner exercises the tasks of equal point count (quantile segments,
    empty tasks).  ComplexF64 at M = 10 has no ring (no LDS tile of 4-cell bins fits 160 KiB, the plan bins by 2 cells and the ring
    needs bins of 4): LDS tiles there, and for the plan with the ring switched off at the end."""
    dims, Np = (48, 40, 56), 4000
    monkeypatch.setenv("NUFFT_INTERP_MARCH", "2")
    # ComplexF64: part by part through the real kernel (the default), and the complex instantiations (NUFFT_INTERP_SPLIT=0)
    for split in (("1", "0") if np.dtype(Z) == np.complex128 else ("1",)):
        monkeypatch.setenv("NUFFT_INTERP_SPLIT", split)
        has_ring = not (np.dtype(Z) == np.complex128 and M >= 10)
        for evalmode in (O.FAST_APPROXIMATION, O.DIRECT):
            nufft, plan, oplan, xs, vs = _make_case(Z, dims, M, 2.0, evalmode, 1, Np, seed=100 + M)
            dev = plan.device
            rng = np.random.default_rng(5 + M)
            w = (rng.standard_normal(plan.shape) + 1j * rng.standard_normal(plan.shape))
            w = w.astype(np.complex64 if plan_real_dtype(Z) == np.float32 else np.complex128)
            wd = torch.from_numpy(w).to(dev)
            for name in ("uniform", "corner"):
                pts = xs if name == "uniform" else tuple((0.3 * x * x / (2 * np.pi)).astype(x.dtype) for x in xs)
                nufft.set_points(plan, tuple(torch.from_numpy(x).to(dev) for x in pts))
                O.set_points(oplan, pts)
                out = torch.empty(Np, dtype=plan.Z, device=dev)
                nufft.exec_type2(out, plan, wd)
                assert plan.interp_engine_used() == ("marching_ring" if has_ring else "lds_tiles"), (name, evalmode, split)
                ref = O.exec_type2(oplan, _oracle_inputs(oplan, [w])[0])
                assert _rel(out.cpu().numpy(), ref) < _rtol(Z), (name, evalmode, split)
    # the same transform with the ring switched off: LDS-tile kernel, same result
    monkeypatch.setenv("NUFFT_INTERP_MARCH", "0")
    nufft, plan2, _, _, _ = _make_case(Z, dims, M, 2.0, O.DIRECT, 1, Np, seed=100 + M)
    nufft.set_points(plan2, tuple(torch.from_numpy(x).to(dev) for x in pts))
    out2 = torch.empty(Np, dtype=plan2.Z, device=dev)
    nufft.exec_type2(out2, plan2, wd)
    assert plan2.interp_engine_used() == "lds_tiles"
    assert _rel(out2.cpu().numpy(), ref) < _rtol(Z)


@pytest.mark.parametrize("M,C", [(4, 2), (7, 3)])
def test_interpolation_ring_complex128_components_part_by_part(M, C, monkeypatch):
    """ComplexF64 with ntransforms > 1 through the real ring kernels: a launch of 2 x tasks workgroups per component (the two parts of a
    task side by side, march_setup.inc), type 2 against the oracle per component."""
    monkeypatch.setenv("NUFFT_INTERP_MARCH", "2")
    dims, Np = (48, 40, 56), 5000
    nufft, plan, oplan, xs, vs = _make_case(np.complex128, dims, M, 2.0, O.DIRECT, C, Np, seed=300 + M)
    dev = plan.device
    rng = np.random.default_rng(M)
    ws = [(rng.standard_normal(plan.shape) + 1j * rng.standard_normal(plan.shape)) for _ in range(C)]
    wd = tuple(torch.from_numpy(w).to(dev) for w in ws)
    nufft.set_points(plan, tuple(torch.from_numpy(x).to(dev) for x in xs))
    O.set_points(oplan, xs)
    outs = tuple(torch.empty(Np, dtype=plan.Z, device=dev) for _ in range(C))
    nufft.exec_type2(outs, plan, wd)
    assert plan.interp_engine_used() == "marching_ring"
    ref = O.exec_type2(oplan, ws)
    for c in range(C):
        assert _rel(outs[c].cpu().numpy(), ref[c]) < _rtol(np.complex128), c


@pytest.mark.parametrize("M", range(2, 11))
@pytest.mark.parametrize("Z", [np.float32, np.complex64, np.float64, np.complex128])
def test_spreading_ring_every_instantiation(Z, M, monkeypatch):
    """Every (element type, M) instantiation of spread_march_kernel (clipped columns: the halo variant has its own test below)
    against the oracle (type 1, both window evaluations), the
    ring requested explicitly and nufft_spread_engine_used confirming the device-side flag.  Oversampled grid 96 x 80 x 112:
    partial columns at the upper ends of dimensions 1 and 2, columns at the periodic boundary (two runs per row of bins),
    several segments along dimension 3 (first / last layers clip along z, the others take the per-slot code); a point set
    concentrated in a corner exercises the tasks of equal point count (quantile segments, empty tasks that only store zeros)."""
    dims, Np = (48, 40, 56), 4000
    monkeypatch.setenv("NUFFT_SMARCH_HALO", "0")
    # complex data: part by part through the real kernel (the default), and the interleaved complex instantiations (NUFFT_SMARCH_SPLIT=0)
    cases = [(e, "1") for e in (O.FAST_APPROXIMATION, O.DIRECT)] + ([(O.FAST_APPROXIMATION, "0"), (O.DIRECT, "0")] if np.dtype(Z).kind == "c" else [])
    for evalmode, split in cases:
        monkeypatch.setenv("NUFFT_SMARCH_SPLIT", split)
        if np.dtype(Z) == np.complex128 and M == 10:
            # the LDS tiles of this plan need 2-cell bins (plan_math.cpp); the ring, like the patches, sorts by 4-cell bins
            nufft, plan, *_ = _make_case(Z, dims, M, 2.0, evalmode, 1, Np, seed=300 + M)
            assert plan.info().bin_dims[0] == 2
            with pytest.raises(ValueError):
                _make_case(Z, dims, M, 2.0, evalmode, 1, Np, seed=300 + M, spread_method="marching_ring")
            return
        nufft, plan, oplan, xs, vs = _make_case(Z, dims, M, 2.0, evalmode, 1, Np, seed=300 + M, spread_method="marching_ring")
        assert plan.info().spread_method == 3 and plan.info().ring_column[0] > 0 and plan.info().ring_halo == 0
        dev = plan.device
        for name in ("uniform", "corner"):
            pts = xs if name == "uniform" else tuple((0.3 * x * x / (2 * np.pi)).astype(x.dtype) for x in xs)
            nufft.set_points(plan, tuple(torch.from_numpy(x).to(dev) for x in pts))
            O.set_points(oplan, pts)
            u = torch.empty(plan.shape, dtype=plan.eltype, device=dev)
            nufft.exec_type1(u, plan, torch.from_numpy(vs[0]).to(dev))
            assert plan.spread_engine_used() == "marching_ring", (name, evalmode)
            ref = O.exec_type1(oplan, _oracle_inputs(oplan, vs)[0])
            assert _rel(u.cpu().numpy(), ref) < _rtol(Z), (name, evalmode)


@pytest.mark.parametrize("M", range(2, 11))
@pytest.mark.parametrize("Z,C", [(np.float32, 1), (np.float64, 1), (np.float64, 2), (np.complex64, 1), (np.complex128, 1)])
def test_spreading_ring_halo_variant_every_instantiation(Z, C, M, monkeypatch):
    _ring_halo_variant_case(Z, C, M, monkeypatch, split=True)


@pytest.mark.parametrize("M", range(2, 7))
@pytest.mark.parametrize("Z,C", [(np.float32, 1), (np.float64, 1), (np.float64, 2), (np.complex64, 1), (np.complex128, 2)])
def test_dense_window_engine_every_instantiation(Z, C, M, monkeypatch):
    """The dense-set engine of the spreading window (dmarch_kernels.h, round 6: the points of a 4^3-cell bin accumulated in registers by
    v_mfma_f64_16x16x4, one flush of the bin's (2M + 3)^3 footprint into the window) for every element type and M = 2 .. 6, both window
    evaluations, against the C oracle (Float64; Float32 plans: points located in Float32 as a Float32 plan of the reference does, bound 1e-5):
    a dense uniform set (about 20 points per bin: full and partly filled batches of four, several loads of sixteen records per bin), a set
    concentrated in a corner (bins with hundreds of points next to empty ones; tasks of equal point count with clipped first / last
    layers), and a sparse one that the plan itself would not give to this engine (forced: bins with one to three points, mostly empty
    batch slots).  The engine is read back; set_points picks it from the mean bin load (the dense set: the plan's own threshold).  Grid
    96 x 96 x 112 as for the halo variant above: columns at the periodic boundary, several segments along dimension 3.  Also: the
    stage-level entry point (complete grid against the oracle's spread field) and type 2 on the fine-sorted point set.
    Semantics: src/spreading/gpu.jl:237-377."""
    from oracle import c_oracle as CO
    if not CO.available():
        pytest.skip("C oracle not built")
    nufft = _nufft()
    Zt = np.dtype(Z)
    is_real = Zt.kind == "f"
    T = plan_real_dtype(Z)
    wide = np.float64 if is_real else np.complex128
    tol = 1e-5 if T == np.float32 else 1e-7
    dims = (48, 48, 56)
    nbins = (96 // 4) * (96 // 4) * (112 // 4)
    monkeypatch.setenv("NUFFT_SMARCH_HALO", "2")
    rng = np.random.default_rng(1300 + M)
    for evalmode in (O.DIRECT, O.FAST_APPROXIMATION):
        mode = nufft.Direct() if evalmode == O.DIRECT else nufft.FastApproximation()
        oplan = O.OraclePlan(dims, is_real=is_real, dtype=np.float64, coord_dtype=(np.float32 if T == np.float32 else None), M=M, sigma=2.0,
                             evalmode=evalmode, ntransforms=C)
        for name, Np, force in (("dense", 20 * nbins, False), ("corner", 6 * nbins, True), ("sparse", nbins // 2, True)):
            # (forced for the sets the plan's own rule would leave to the atomic window; the dense set: an explicit threshold of 16 points per bin)
            monkeypatch.setenv("NUFFT_DENSE_MIN", "0" if force else "16")
            plan = nufft.PlanNUFFT(Zt, dims, m=M, sigma=2.0, ntransforms=C, kernel_evalmode=mode, spread_method="marching_ring", backend=nufft.ROCBackend(0))
            info = plan.info()
            assert info.spread_method == 3 and info.ring_halo == 1, (M, list(info.ring_column))
            dev = plan.device
            xs = [((rng.random(Np) * 3 - 1) * O.TWO_PI).astype(T) for _ in dims]      # points outside the unit cell too
            if name == "corner":
                xs = [(0.3 * np.mod(x, T(O.TWO_PI)) ** 2 / (2 * np.pi)).astype(T) for x in xs]
            vs = [(rng.standard_normal(Np) if is_real else rng.standard_normal(Np) + 1j * rng.standard_normal(Np)).astype(Zt) for _ in range(C)]
            nufft.set_points(plan, tuple(torch.from_numpy(x).to(dev) for x in xs))
            O.set_points(oplan, xs)
            us = tuple(torch.empty(plan.shape, dtype=plan.eltype, device=dev) for _ in range(C))
            vd = tuple(torch.from_numpy(v).to(dev) for v in vs)
            nufft.exec_type1(us if C > 1 else us[0], plan, vd if C > 1 else vd[0])
            assert plan.spread_engine_used() == "marching_ring_dense", (name, evalmode, M)
            ref = CO.exec_type1(oplan, [v.astype(wide) for v in vs])
            for c in range(C):
                assert _rel(us[c].cpu().numpy().astype(np.complex128), ref[c]) < tol, ("type 1", name, evalmode, c)
            if name == "dense" and evalmode == O.DIRECT:
                refg = CO.spread(oplan, [v.astype(wide) for v in vs])
                nufft.spread_from_points(plan, vd if C > 1 else vd[0])
                scale = 2.0 ** sum(info.window_scale_log2[d] for d in range(3))
                for c in range(C):
                    gc = nufft.oversampled_grid(plan, c).cpu().numpy().astype(wide) / scale
                    assert _rel(gc, refg[c]) < (1e-12 if T == np.float64 else 2e-6), ("grid", c)
                ctype = np.complex64 if T == np.float32 else np.complex128
                ws = [(rng.standard_normal(plan.shape) + 1j * rng.standard_normal(plan.shape)).astype(ctype) for _ in range(C)]
                out = tuple(torch.empty(Np, dtype=vd[0].dtype, device=dev) for _ in range(C))
                wd = tuple(torch.from_numpy(w).to(dev) for w in ws)
                nufft.exec_type2(out if C > 1 else out[0], plan, wd if C > 1 else wd[0])
                ref2 = CO.exec_type2(oplan, [w.astype(np.complex128) for w in ws])
                for c in range(C):
                    assert _rel(out[c].cpu().numpy().astype(wide), ref2[c]) < tol, ("type 2", c)
                # the plan's own rule (no switch): measured break-even per (M, window) — at ~19 points per bin m = 5, 6 take the engine, m <= 4 with
                # Direct() do not (plan.cpp: dense_min_direct / dense_min_poly)
                monkeypatch.delenv("NUFFT_DENSE_MIN", raising=False)
                own = nufft.PlanNUFFT(Zt, dims, m=M, sigma=2.0, ntransforms=C, kernel_evalmode=mode, spread_method="marching_ring", backend=nufft.ROCBackend(0))
                nufft.set_points(own, tuple(torch.from_numpy(x).to(dev) for x in xs))
                assert own.spread_engine_used() == ("marching_ring_dense" if M >= 5 else "marching_ring"), M
                own.close()
            plan.close()


@pytest.mark.parametrize("M", range(2, 6))
@pytest.mark.parametrize("Z", [np.complex64, np.complex128])
def test_spreading_ring_halo_variant_interleaved_complex_instantiations(Z, M, monkeypatch):
    """The complex instantiations of the halo variant (interleaved window; NUFFT_SMARCH_SPLIT=0) — by default complex data runs part by
    part through the real kernel (the test above)."""
    _ring_halo_variant_case(Z, 1, M, monkeypatch, split=False)


def _ring_halo_variant_case(Z, C, M, monkeypatch, split):
    """The halo variant of spread_march_kernel (every point spread once by its own column, the stencil's reach beyond
    the column through a side buffer) for every M, both window evaluations, against the oracle — three consumers of the side
    buffer: the plan's own dimension-1 FFT pass (exec_type1), the separate add pass in front of the FFT
    (NUFFT_SMARCH_HALO_FUSE=0), and the stage-level entry point (spread_from_points: the oversampled grid itself against the
    oracle's).  Oversampled grid 96 x 96 x 112: whole columns (the variant needs them), columns at the periodic boundary whose
    reach wraps around, several segments along dimension 3, and a point set concentrated in a corner (tasks of equal point count).
    Where no halo kernel exists (LDS: Float64 at M >= 8, complex data at M >= 6) the plan falls back to the clipped columns —
    the variant is asserted for M <= 7 (real) / M <= 5 (complex)."""
    dims, Np = (48, 48, 56), 4000
    is_real = np.dtype(Z).kind == "f"
    if np.dtype(Z) == np.complex128 and M == 10:
        pytest.skip("2-cell bins: no ring (test_spreading_ring_every_instantiation)")
    monkeypatch.setenv("NUFFT_SMARCH_HALO", "2")
    monkeypatch.setenv("NUFFT_SMARCH_SPLIT", "1" if split else "0")
    for evalmode, fuse in ((O.FAST_APPROXIMATION, "1"), (O.DIRECT, "1"), (O.FAST_APPROXIMATION, "0")):
        monkeypatch.setenv("NUFFT_SMARCH_HALO_FUSE", fuse)
        nufft, plan, oplan, xs, vs = _make_case(Z, dims, M, 2.0, evalmode, C, Np, seed=500 + M, spread_method="marching_ring")
        info = plan.info()
        assert info.spread_method == 3 and info.ring_column[0] > 0
        assert 96 % info.ring_column[0] == 0 and 96 % info.ring_column[1] == 0 or info.ring_halo == 0
        if M <= (7 if (is_real or split) and not (np.dtype(Z) in (np.dtype(np.float64), np.dtype(np.complex128)) and M > 7) else 5):
            assert info.ring_halo == 1, (M, list(info.ring_column))
        dev = plan.device
        for name in ("uniform", "corner"):
            pts = xs if name == "uniform" else tuple((0.3 * x * x / (2 * np.pi)).astype(x.dtype) for x in xs)
            nufft.set_points(plan, tuple(torch.from_numpy(x).to(dev) for x in pts))
            O.set_points(oplan, pts)
            us = tuple(torch.empty(plan.shape, dtype=plan.eltype, device=dev) for _ in range(C))
            vd = tuple(torch.from_numpy(v).to(dev) for v in vs)
            nufft.exec_type1(us if C > 1 else us[0], plan, vd if C > 1 else vd[0])
            assert plan.spread_engine_used() == "marching_ring", (name, evalmode)
            ref = O.exec_type1(oplan, _oracle_inputs(oplan, vs) if C > 1 else _oracle_inputs(oplan, vs)[0])
            for c in range(C):
                assert _rel(us[c].cpu().numpy(), ref[c] if C > 1 else ref) < _rtol(Z), (name, evalmode, fuse, c)
        if fuse == "1" and evalmode == O.DIRECT:
            # stage level: the grid after spread_from_points is complete (the add pass ran) — against the oracle's grid
            nufft.spread_from_points(plan, vd if C > 1 else vd[0])
            scale = 2.0 ** sum(info.window_scale_log2[d] for d in range(3))
            wide = np.float64 if is_real else np.complex128
            o64 = O.OraclePlan(dims, is_real=is_real, dtype=np.float64, M=M, sigma=2.0, evalmode=evalmode, ntransforms=C)
            O.set_points(o64, [x.astype(np.float64) for x in pts])
            refg = O.spread(o64, [v.astype(wide) for v in vs])
            for c in range(C):
                grid = nufft.oversampled_grid(plan, c).cpu().numpy().astype(wide) / scale
                assert _rel(grid, refg[c]) < (1e-12 if np.dtype(Z) in (np.float64, np.complex128) else 1e-5), (M, c)


@pytest.mark.parametrize("Z,C,fuse", [(np.float64, 1, "1"), (np.float64, 2, "1"), (np.float32, 1, "1"), (np.float64, 1, "0")])
def test_oversampled_grid_after_exec_type1_on_halo_plans(Z, C, fuse, monkeypatch):
    """`us` after exec_type1! is the spread field (the reference's r2c is out of place: `mul!(ûs, plan_fw, us)`,
    src/NonuniformFFTs.jl:197-204).  On plans of the window's halo variant the fused FFT pass adds the side buffer to the lines it
    loads, not to `us`: the grid is completed on demand (nufft_complete_grid / nufft_copy_grid / nufft_interpolate), once, and
    nufft_grid_ptr refuses while that is pending."""
    import ctypes as Ct
    dims, Np, M = (64, 48, 64), 5000, 4          # 128 x 96 x 128: the plan's own FFT passes, dimension 1 included (the fused consumer exists)
    monkeypatch.setenv("NUFFT_SMARCH_HALO", "2")
    monkeypatch.setenv("NUFFT_SMARCH_HALO_FUSE", fuse)
    nufft, plan, oplan, xs, vs = _make_case(Z, dims, M, 2.0, O.DIRECT, C, Np, seed=4242, spread_method="marching_ring")
    info = plan.info()
    assert info.spread_method == 3 and info.ring_halo == 1
    dev = plan.device
    nufft.set_points(plan, tuple(torch.from_numpy(x).to(dev) for x in xs))
    us = tuple(torch.empty(plan.shape, dtype=plan.eltype, device=dev) for _ in range(C))
    vd = tuple(torch.from_numpy(v).to(dev) for v in vs)
    o64 = O.OraclePlan(dims, is_real=True, dtype=np.float64, M=M, sigma=2.0, evalmode=O.DIRECT, ntransforms=C)
    O.set_points(o64, [x.astype(np.float64) for x in xs])
    refg = O.spread(o64, [v.astype(np.float64) for v in vs])
    scale = 2.0 ** sum(info.window_scale_log2[d] for d in range(3))
    tol = 1e-12 if np.dtype(Z) == np.float64 else 1e-5
    lib, h, stream = nufft.lib, plan._handle, plan._stream()
    ptr, nbytes = Ct.c_void_p(), Ct.c_int64()

    nufft.exec_type1(us if C > 1 else us[0], plan, vd if C > 1 else vd[0])
    pending = fuse == "1"          # (the separate add pass in front of the FFT leaves nothing pending)
    rc = lib.nufft_grid_ptr(h, 0, 0, Ct.byref(ptr), Ct.byref(nbytes))
    assert rc == (nufft._lib.ERR_INVALID_ARG if pending else 0)
    assert lib.nufft_grid_ptr(h, 1, 0, Ct.byref(ptr), Ct.byref(nbytes)) == 0          # the spectrum is not affected
    grids = [nufft.oversampled_grid(plan, c).cpu().numpy().astype(np.float64) / scale for c in range(C)]      # nufft_copy_grid completes
    for c in range(C):
        assert _rel(grids[c], refg[c]) < tol, c
    assert lib.nufft_grid_ptr(h, 0, 0, Ct.byref(ptr), Ct.byref(nbytes)) == 0
    again = nufft.oversampled_grid(plan, 0).cpu().numpy().astype(np.float64) / scale                           # ... once: not added twice
    assert np.array_equal(again, grids[0])

    # nufft_complete_grid, then the raw pointer; and a second forward FFT + deconvolution on the completed grid gives the same modes
    nufft.exec_type1(us if C > 1 else us[0], plan, vd if C > 1 else vd[0])
    first = [u.clone() for u in us]
    assert lib.nufft_complete_grid(h, stream) == 0 and lib.nufft_complete_grid(h, stream) == 0
    assert lib.nufft_grid_ptr(h, 0, 0, Ct.byref(ptr), Ct.byref(nbytes)) == 0 and nbytes.value == grids[0].size * np.dtype(Z).itemsize
    assert lib.nufft_fft_forward(h, stream) == 0
    assert lib.nufft_deconvolve_truncate(h, nufft.plan._ptr_table(us), stream) == 0
    for c in range(C):
        assert _rel(us[c].cpu().numpy(), first[c].cpu().numpy()) < 10 * tol

    # stage level: interpolation from the grid an exec_type1 left behind = interpolation of the oracle's spread field
    nufft.exec_type1(us if C > 1 else us[0], plan, vd if C > 1 else vd[0])
    out = tuple(torch.empty(Np, dtype=vd[0].dtype, device=dev) for _ in range(C))
    nufft.interpolate(plan, out if C > 1 else out[0])
    refv = O.interpolate(o64, [g.copy() for g in refg])
    for c in range(C):
        assert _rel(out[c].cpu().numpy().astype(np.float64) / scale ** 2, refv[c]) < (tol if np.dtype(Z) == np.float64 else 2e-5), c


@pytest.mark.parametrize("Z,dims,M,sigma,kname", [(np.float64, (31, 20, 17), 4, 1.5, "BackwardsKaiserBesselKernel"),
                                                 (np.float32, (35, 24), 5, 1.25, "KaiserBesselKernel"),
                                                 (np.complex128, (33, 20), 6, 2.0, "GaussianKernel"),
                                                 (np.float64, (64, 64, 64), 4, 2.0, "BackwardsKaiserBesselKernel")])
def test_forwarded_kernel_data_reproduces_the_plan(Z, dims, M, sigma, kname):
    """What the Julia extension sends (julia/ext/NonuniformFFTsMI355XExt.jl `ensure_handle!`): the oversampled sizes and the shape
    parameter per dimension as the reference's plan holds them (`N_over`, `kernel_param_dim`), and N1 = 2 (L - 1) for real data
    whatever the parity of N1.  The plan built from them transforms exactly like the plan built from (Ns, σ, kernel)."""
    nufft = _nufft()
    Z = np.dtype(Z)
    D = len(dims)
    kcls = getattr(nufft, kname)
    p = nufft.PlanNUFFT(Z, dims, m=M, sigma=sigma, kernel=kcls(), backend=nufft.ROCBackend(0))
    info = p.info()
    Ls = [int(info.N_out[d]) for d in range(D)]
    sent = tuple((2 * (Ls[0] - 1) if (Z.kind == "f" and d == 0) else Ls[d]) for d in range(D))
    q = nufft.PlanNUFFT(Z, sent, m=M, sigma=float(info.sigma), kernel=kcls(), backend=nufft.ROCBackend(0),
                        kernel_param_dim=[info.beta[d] for d in range(D)], oversampled_dims=[int(info.N_over[d]) for d in range(D)])
    assert q.shape == p.shape and q.oversampled_dims == p.oversampled_dims
    rng = np.random.default_rng(7)
    Np = 3000
    T = plan_real_dtype(Z)
    xs = tuple(torch.from_numpy((rng.random(Np) * O.TWO_PI).astype(T)).cuda() for _ in dims)
    v = rng.standard_normal(Np) + (1j * rng.standard_normal(Np) if Z.kind == "c" else 0)
    v = torch.from_numpy(v.astype(Z)).cuda()
    outs = []
    for plan in (p, q):
        nufft.set_points(plan, xs)
        u = torch.empty(plan.shape, dtype=plan.eltype, device="cuda")
        nufft.exec_type1(u, plan, v)
        w = torch.empty(Np, dtype=v.dtype, device="cuda")
        nufft.exec_type2(w, plan, u)
        outs.append((u.cpu().numpy(), w.cpu().numpy()))
    # same plan constants bit for bit (tests/test_host_abi.py): only the summation order of the atomics differs between two runs —
    # Float32: 1e-6 on the grid, amplified by the deconvolution at sigma = 1.25 (the reference's own Float32 criterion is 1e-5)
    tol = 1e-13 if T == np.float64 else 5e-5
    assert _rel(outs[1][0], outs[0][0]) < tol and _rel(outs[1][1], outs[0][1]) < tol


# every (element type, M) for which the spreading window runs in its halo variant: since round 6 the interpolation ring takes the window's
# column (one chooser, plan.cpp shared_ring_column) and interp_march_staged_kernel is instantiated for all of them — the list holds only
# eligible combinations and the test ASSERTS eligibility (tests/test_host_abi.py::test_gpu_parametrisations_are_eligible_by_construction
# checks the same from host-only plans, so a list that would mostly skip fails on the CPU already)
COLUMN_LAYER_CASES = [(np.float64, 4, 1, O.FAST_APPROXIMATION), (np.float64, 4, 1, O.DIRECT), (np.float64, 4, 3, O.FAST_APPROXIMATION),
                      (np.float32, 4, 1, O.DIRECT), (np.float64, 2, 1, O.DIRECT), (np.float64, 3, 2, O.FAST_APPROXIMATION),
                      (np.float64, 5, 1, O.DIRECT), (np.float32, 6, 1, O.FAST_APPROXIMATION), (np.float64, 6, 1, O.DIRECT),
                      (np.float32, 7, 1, O.DIRECT), (np.float32, 3, 1, O.DIRECT), (np.float32, 3, 1, O.FAST_APPROXIMATION), (np.float32, 5, 2, O.DIRECT),
                      (np.float64, 6, 1, O.FAST_APPROXIMATION), (np.float64, 5, 1, O.FAST_APPROXIMATION),
                      # ComplexF64: both rings run the real kernels part by part; ComplexF32: the window part by part, the paired-lane ring
                      (np.complex128, 4, 1, O.FAST_APPROXIMATION), (np.complex128, 4, 2, O.DIRECT), (np.complex128, 6, 2, O.DIRECT),
                      (np.complex128, 2, 1, O.DIRECT), (np.complex128, 5, 1, O.FAST_APPROXIMATION),
                      (np.complex64, 4, 1, O.DIRECT), (np.complex64, 6, 1, O.FAST_APPROXIMATION), (np.complex64, 3, 2, O.DIRECT), (np.complex64, 5, 1, O.DIRECT)]
COLUMN_LAYER_DIMS = (256, 256, 32)          # 512 x 512 x 64: 16 x 16 columns of 32 x 32 cells as at C2, 16 layers of bins


@pytest.mark.parametrize("Z,M,C,evalmode", COLUMN_LAYER_CASES + [(np.float32, 2, 1, O.FAST_APPROXIMATION), (np.float32, 2, 1, O.DIRECT)])
def test_column_layer_sort_and_staged_interpolation(Z, M, C, evalmode, monkeypatch):
    """Plans whose spreading window (halo variant) and interpolation ring own the same columns sort the points by (column, layer of
    bins) only (binsort.hip, CoarseSort: LDS histograms, no global atomics) and interpolate with interp_march_staged_kernel, which
    orders a layer's points by bin on their way into LDS.  Type 1 and type 2 against the oracle on a uniform set (column-layer sort,
    read back), with per-point weights (the ring applies them itself).  The sort result is a permutation grouped by column layer."""
    dims, Np = COLUMN_LAYER_DIMS, 60000
    monkeypatch.setenv("NUFFT_SMARCH_HALO", "2")
    monkeypatch.delenv("NUFFT_COARSE_SORT", raising=False)
    if np.dtype(Z) == np.float32 and M == 2:
        # the one plan that keeps its own (64 x 64) ring column by default — measured slower with the shared one (plan.cpp shared_ring_column):
        # its staged instantiation is tested with the column forced
        monkeypatch.setenv("NUFFT_COARSE_SORT", "2")
    monkeypatch.setenv("NUFFT_INTERP_MARCH", "2")                # always the ring: the grid is too small to fill the chip
    nufft, plan, oplan, xs, vs = _make_case(Z, dims, M, 2.0, evalmode, C, Np, seed=900 + M, spread_method="marching_ring")
    info = plan.info()
    assert info.spread_method == 3 and info.ring_halo == 1
    # eligible by construction — and the column is the spreading window's
    assert info.sort_column[0] > 0 and [4 * info.sort_column[0], 4 * info.sort_column[1]] == list(info.ring_column), (list(info.sort_column), list(info.ring_column))
    # ... as the host-only prediction says (what the CPU-side eligibility test relies on)
    hp = nufft.PlanNUFFT(Z, dims, m=M, sigma=2.0, ntransforms=C, kernel_evalmode=nufft.Direct() if evalmode == O.DIRECT else nufft.FastApproximation(),
                         spread_method="marching_ring", backend=None)
    assert list(hp.info().sort_column) == list(info.sort_column)
    dev = plan.device
    tup = (lambda t: t if C > 1 else t[0])
    xd = tuple(torch.from_numpy(x).to(dev) for x in xs)
    vd = tuple(torch.from_numpy(v).to(dev) for v in vs)
    nufft.set_points(plan, xd)
    O.set_points(oplan, xs)
    assert plan.sort_columns_used() and plan.spread_engine_used() == "marching_ring" and plan.interp_engine_used() == "marching_ring"
    # the sorted array: a permutation; every column layer is one contiguous run (its points sit in its first bin's range)
    perm, offs = nufft.sort_result(plan)
    assert np.array_equal(np.sort(perm), np.arange(Np))
    Nover = plan.oversampled_dims
    T = plan_real_dtype(Z)
    twopi = T(O.TWO_PI)
    folded = [np.mod(x, twopi) for x in xs]
    cells = [np.minimum(((f / twopi) * T(n)).astype(np.int64), n - 1) for f, n in zip(folded, Nover)]
    nb = [n // 4 for n in Nover]
    cbx, cby = info.sort_column[0], info.sort_column[1]
    rep = ((cells[2] // 4) * nb[1] + (cells[1] // 4) // cby * cby) * nb[0] + (cells[0] // 4) // cbx * cbx      # first bin of the point's column layer
    pos = np.empty(Np, dtype=np.int64)
    pos[perm] = np.arange(Np)
    offs = offs.astype(np.int64)
    inside = (pos >= offs[rep]) & (pos < offs[rep + 1])
    assert inside.mean() > 0.999        # (points within rounding of a cell boundary may be binned one cell off by this numpy restatement)
    us = tuple(torch.empty(plan.shape, dtype=plan.eltype, device=dev) for _ in range(C))
    nufft.exec_type1(tup(us), plan, tup(vd))
    ref = O.exec_type1(oplan, _oracle_inputs(oplan, vs))
    for c in range(C):
        assert _rel(us[c].cpu().numpy(), ref[c]) < _rtol(Z), ("type 1", c)
    rng = np.random.default_rng(5)
    ctype = np.complex64 if T == np.float32 else np.complex128
    ws = [(rng.standard_normal(plan.shape) + 1j * rng.standard_normal(plan.shape)).astype(ctype) for _ in range(C)]
    wd = tuple(torch.from_numpy(w).to(dev) for w in ws)
    out = tuple(torch.empty(Np, dtype=vd[0].dtype, device=dev) for _ in range(C))
    nufft.exec_type2(tup(out), plan, tup(wd))
    ref2 = O.exec_type2(oplan, _oracle_inputs(oplan, ws))
    for c in range(C):
        assert _rel(out[c].cpu().numpy(), ref2[c]) < _rtol(Z), ("type 2", c)
    # callbacks menu on the same point set: per-point weights in both directions (the staged ring multiplies them in)
    wts = (rng.random(Np) + 0.5).astype(T)
    cb = nufft.NUFFTCallbacks(nonuniform=nufft.PointWeights(torch.from_numpy(wts).to(dev)))
    ocb = O.NUFFTCallbacks(nonuniform=lambda v, n: tuple(type(x)(x * wts[n]) for x in v))
    nufft.exec_type1(tup(us), plan, tup(vd), callbacks=cb)
    ref = O.exec_type1(oplan, _oracle_inputs(oplan, vs), callbacks=ocb)
    for c in range(C):
        assert _rel(us[c].cpu().numpy(), ref[c]) < _rtol(Z), ("type 1 with weights", c)
    nufft.exec_type2(tup(out), plan, tup(wd), callbacks=cb)
    ref2 = O.exec_type2(oplan, _oracle_inputs(oplan, ws), callbacks=ocb)
    for c in range(C):
        assert _rel(out[c].cpu().numpy(), ref2[c]) < _rtol(Z), ("type 2 with weights", c)


def test_column_layer_sort_falls_back_to_fine_bins_for_clustered_points(monkeypatch):
    """The decision is per point set, on the device: a set that either ring hands to the tile kernels is sorted by fine bins (the tile
    kernels need them), the next uniform set by column layers again — same plan, no host read-back in between."""
    for var in ("NUFFT_INTERP_MARCH", "NUFFT_SPREAD_METHOD", "NUFFT_PREFER_RING", "NUFFT_PREFER_PATCHES", "NUFFT_SMARCH_ADVANTAGE", "NUFFT_COARSE_SORT",
                "NUFFT_SMARCH_HALO"):
        monkeypatch.delenv(var, raising=False)
    dims, Np = (256, 256, 32), 120000       # 512 x 512 x 64: the columns of C2, enough of them for the automatic choice to take the rings
    nufft, plan, oplan, xs, vs = _make_case(np.float64, dims, 4, 2.0, O.FAST_APPROXIMATION, 1, Np, seed=78)
    info = plan.info()
    assert info.spread_method == 3 and info.ring_halo == 1 and info.sort_column[0] > 0, list(info.ring_column)
    dev = plan.device
    u = torch.empty(plan.shape, dtype=plan.eltype, device=dev)
    out = torch.empty(Np, dtype=torch.float64, device=dev)
    rng = np.random.default_rng(6)
    w = rng.standard_normal(plan.shape) + 1j * rng.standard_normal(plan.shape)
    for name in ("uniform", "corner", "uniform"):
        pts = xs if name == "uniform" else tuple((0.05 * x).astype(x.dtype) for x in xs)
        nufft.set_points(plan, tuple(torch.from_numpy(x).to(dev) for x in pts))
        O.set_points(oplan, pts)
        assert plan.sort_columns_used() == (name == "uniform"), name
        assert plan.spread_engine_used() == ("marching_ring" if name == "uniform" else "lds_tiles"), name
        nufft.exec_type1(u, plan, torch.from_numpy(vs[0]).to(dev))
        assert _rel(u.cpu().numpy(), O.exec_type1(oplan, vs[0])) < 1e-7, name
        nufft.exec_type2(out, plan, torch.from_numpy(w).to(dev))
        assert _rel(out.cpu().numpy(), O.exec_type2(oplan, w)) < 1e-7, name
    # and with the switch off the same plan parameters never sort by column layers
    monkeypatch.setenv("NUFFT_COARSE_SORT", "0")
    q = nufft.PlanNUFFT(np.float64, dims, m=4, sigma=2.0, backend=nufft.ROCBackend(0))
    assert q.info().sort_column[0] == 0


def test_adaptive_sort_choice_follows_the_rings_decisions(monkeypatch):
    """Plans of the column-layer sort (round 6: most 3-D plans): a point set that a ring hands to the tile kernels ends its column-layer attempt in
    the fine sort with global atomics (1.8 ms at 1.7e7 folded-normal points against 1.1 ms for the slab sort).  The rings' decisions come back to
    the host through host-mapped memory (no synchronisation); after TWO such sets in a row set_points takes the slab sort directly, and goes back
    to the column-layer sort once both rings would serve a set again (plan.cpp, nufft_internal.h: sort_feedback).  Every set, whichever sort it
    took, transforms correctly (types 1 and 2 against the oracle)."""
    for var in ("NUFFT_INTERP_MARCH", "NUFFT_SPREAD_METHOD", "NUFFT_PREFER_RING", "NUFFT_PREFER_PATCHES", "NUFFT_SMARCH_ADVANTAGE", "NUFFT_COARSE_SORT",
                "NUFFT_SMARCH_HALO", "NUFFT_SORT_ADAPTIVE"):
        monkeypatch.delenv(var, raising=False)
    dims, Np = (256, 256, 32), 120000
    nufft, plan, oplan, xs, vs = _make_case(np.float64, dims, 4, 2.0, O.DIRECT, 1, Np, seed=79)
    assert plan.info().sort_column[0] > 0
    dev = plan.device
    u = torch.empty(plan.shape, dtype=plan.eltype, device=dev)
    out = torch.empty(Np, dtype=torch.float64, device=dev)
    rng = np.random.default_rng(8)
    w = rng.standard_normal(plan.shape) + 1j * rng.standard_normal(plan.shape)
    # (a quarter of every axis: a ring declines the set, and its fullest slab still fits the slab sort's second level)
    corner = tuple((0.25 * np.mod(x, 2 * np.pi)).astype(x.dtype) for x in xs)
    #            point set   sort the host enqueues / the device ends up with
    script = [("uniform", xs, "column_layers"),
              ("corner", corner, "fine_bins"),       # column-layer attempt, the device falls back (first miss)
              ("corner", corner, "fine_bins"),       # second miss in a row ...
              ("corner", corner, "slabs"),           # ... now the slab sort directly
              ("corner", corner, "slabs"),
              ("uniform", xs, "slabs"),              # still the slab sort (the host cannot know), but both rings serve the set:
              ("uniform", xs, "column_layers"),      # back to the column-layer sort
              ("corner", corner, "fine_bins")]       # and a single clustered set does not switch
    for step, (name, pts, want) in enumerate(script):
        nufft.set_points(plan, tuple(torch.from_numpy(x).to(dev) for x in pts))
        torch.cuda.synchronize()                     # (the feedback record of this set has arrived before the next set_points looks)
        assert plan.sort_method_used() == want, (step, name, plan.sort_method_used())
        O.set_points(oplan, pts)
        nufft.exec_type1(u, plan, torch.from_numpy(vs[0]).to(dev))
        assert _rel(u.cpu().numpy(), O.exec_type1(oplan, vs[0])) < 1e-7, (step, name)
        nufft.exec_type2(out, plan, torch.from_numpy(w).to(dev))
        assert _rel(out.cpu().numpy(), O.exec_type2(oplan, w)) < 1e-7, (step, name)
    # with the switch off the fallback stays the fine sort
    monkeypatch.setenv("NUFFT_SORT_ADAPTIVE", "0")
    q = nufft.PlanNUFFT(np.float64, dims, m=4, sigma=2.0, backend=nufft.ROCBackend(0))
    for _ in range(4):
        nufft.set_points(q, tuple(torch.from_numpy(x).to(dev) for x in corner))
        torch.cuda.synchronize()
        assert q.sort_method_used() == "fine_bins"


def test_halo_side_buffer_allocation_failure_keeps_the_ring(monkeypatch):
    """when the side buffer (half a grid per component) cannot be allocated the plan keeps the ring with clipped
    columns instead of failing (NUFFT_TEST_HALO_ALLOC_FAIL simulates the failure)."""
    monkeypatch.setenv("NUFFT_TEST_HALO_ALLOC_FAIL", "1")
    dims, Np = (48, 48, 56), 3000
    nufft, plan, oplan, xs, vs = _make_case(np.float64, dims, 4, 2.0, O.DIRECT, 1, Np, seed=99, spread_method="marching_ring")
    assert plan.info().spread_method == 3 and plan.info().ring_halo == 0
    dev = plan.device
    nufft.set_points(plan, tuple(torch.from_numpy(x).to(dev) for x in xs))
    O.set_points(oplan, xs)
    u = torch.empty(plan.shape, dtype=plan.eltype, device=dev)
    nufft.exec_type1(u, plan, torch.from_numpy(vs[0]).to(dev))
    assert plan.spread_engine_used() == "marching_ring"
    assert _rel(u.cpu().numpy(), O.exec_type1(oplan, vs[0])) < 1e-7


@pytest.mark.parametrize("Z,C", [(np.float64, 1), (np.float64, 3), (np.float32, 2), (np.complex128, 1)])
def test_spreading_ring_automatic_choice_and_fallback(Z, C, monkeypatch):
    """Automatic engine choice on a grid with enough columns for the chip (256 x 256 x 64 oversampled): real plans at M = 4
    and ComplexF64 plans take the ring for a uniform point set and hand a point set concentrated in one corner to the LDS
    tiles (device-side decision, read back); both against the oracle."""
    dims, Np = (128, 128, 32), 60000
    for var in ("NUFFT_INTERP_MARCH", "NUFFT_SPREAD_METHOD", "NUFFT_PREFER_RING", "NUFFT_PREFER_PATCHES", "NUFFT_SMARCH_ADVANTAGE"):
        monkeypatch.delenv(var, raising=False)      # the test is about the automatic choices
    nufft, plan, oplan, xs, vs = _make_case(Z, dims, 4, 2.0, O.FAST_APPROXIMATION, C, Np, seed=77)
    dev = plan.device
    is_ring = plan.info().spread_method == 3
    assert is_ring
    for name in ("uniform", "corner"):
        pts = xs if name == "uniform" else tuple((0.05 * x).astype(x.dtype) for x in xs)
        nufft.set_points(plan, tuple(torch.from_numpy(x).to(dev) for x in pts))
        O.set_points(oplan, pts)
        us = tuple(torch.empty(plan.shape, dtype=plan.eltype, device=dev) for _ in range(C))
        nufft.exec_type1(us if C > 1 else us[0], plan, tuple(torch.from_numpy(v).to(dev) for v in vs) if C > 1 else torch.from_numpy(vs[0]).to(dev))
        if is_ring:
            assert plan.spread_engine_used() == ("marching_ring" if name == "uniform" else "lds_tiles"), name
        ref = O.exec_type1(oplan, _oracle_inputs(oplan, vs) if C > 1 else _oracle_inputs(oplan, vs)[0])
        for c in range(C):
            assert _rel(us[c].cpu().numpy(), ref[c] if C > 1 else ref) < _rtol(Z), (name, c)
        # type 2 on the same point sets: the interpolation ring's own device-side decision — the corner set is 8000 times denser
        # than the grid average where its points are (density veto, balance.hip) and goes to the LDS tiles
        outs = tuple(torch.empty(Np, dtype=plan.Z, device=dev) for _ in range(C))
        nufft.exec_type2(outs if C > 1 else outs[0], plan, us if C > 1 else us[0])
        if name == "corner":
            assert plan.interp_engine_used() == "lds_tiles"
        ref2 = O.exec_type2(oplan, [u.cpu().numpy() for u in us] if C > 1 else _oracle_inputs(oplan, [us[0].cpu().numpy()])[0])
        for c in range(C):
            assert _rel(outs[c].cpu().numpy(), ref2[c] if C > 1 else ref2) < _rtol(Z), (name, c, "type 2")


@pytest.mark.parametrize("Z,M", [(np.float64, 4), (np.complex64, 8), (np.float32, 6), (np.complex128, 5), (np.float64, 8)])
def test_tasks_of_equal_point_count_on_nonuniform_sets(Z, M, monkeypatch):
    """Patch engine and interpolation ring on point sets far from uniform (set_points cuts their columns into segments of
    about equal point count; 72 bin layers along dimension 3: the ring's segments are capped at the 64 layers its run
    tables hold), both engines forced, against the oracle."""
    monkeypatch.setenv("NUFFT_INTERP_MARCH", "2")
    dims, Np = (48, 40, 144), 6000
    nufft, plan, oplan, xs, vs = _make_case(Z, dims, M, 2.0, O.FAST_APPROXIMATION, 1, Np, seed=400 + M, spread_method="mfma_patches")
    dev = plan.device
    rng = np.random.default_rng(9)
    w = (rng.standard_normal(plan.shape) + 1j * rng.standard_normal(plan.shape))
    w = w.astype(np.complex64 if plan_real_dtype(Z) == np.float32 else np.complex128)
    sets = {
        "folded normal": tuple(np.mod(rng.standard_normal(Np), 2 * np.pi).astype(x.dtype) for x in xs),
        "slab": (xs[0], xs[1], (np.pi + 0.15 * rng.standard_normal(Np)).astype(xs[2].dtype)),        # a few bin layers hold everything
        "two columns": ((0.2 * xs[0]).astype(xs[0].dtype), (5.5 + 0.1 * xs[1]).astype(xs[1].dtype), xs[2]),
    }
    for name, pts in sets.items():
        nufft.set_points(plan, tuple(torch.from_numpy(x).to(dev) for x in pts))
        O.set_points(oplan, pts)
        u = torch.empty(plan.shape, dtype=plan.eltype, device=dev)
        nufft.exec_type1(u, plan, torch.from_numpy(vs[0]).to(dev))
        assert plan.spread_engine_used() == "mfma_patches", name
        assert _rel(u.cpu().numpy(), O.exec_type1(oplan, _oracle_inputs(oplan, vs)[0])) < _rtol(Z), name
        out = torch.empty(Np, dtype=plan.Z, device=dev)
        nufft.exec_type2(out, plan, torch.from_numpy(w).to(dev))
        assert plan.interp_engine_used() == "marching_ring", name
        assert _rel(out.cpu().numpy(), O.exec_type2(oplan, _oracle_inputs(oplan, [w])[0])) < _rtol(Z), name


def test_spreading_engine_selection():
    nufft = _nufft()
    # explicit request on a plan the patches cannot serve (2-D; odd oversampled size) -> ArgumentError, nothing silent
    with pytest.raises(ValueError):
        nufft.PlanNUFFT(np.float64, (64, 64), spread_method="mfma_patches", backend=nufft.ROCBackend(0))
    with pytest.raises(ValueError):
        nufft.PlanNUFFT(np.complex128, (35, 64, 40), sigma=1.5, spread_method="mfma_patches", backend=nufft.ROCBackend(0))
    # automatic choice where all three engines are eligible: the marching ring for real data up to
    # M = 6, ComplexF64 up to M = 4 and ComplexF32 up to M = 3; the patches above that; LDS tiles where neither applies
    def method(Z, dims=(64, 64, 64), **kw):
        return nufft.PlanNUFFT(Z, dims, backend=nufft.ROCBackend(0), **kw).info().spread_method
    assert method(np.float64) == 3 and method(np.float64, m=6) == 3 and method(np.float64, m=7) == 2
    # complex data runs part by part through the real window kernel: the window up to M = 6, the patches above
    assert method(np.complex128) == 3 and method(np.complex128, m=5) == 3 and method(np.complex128, m=6) == 3 and method(np.complex128, m=7) == 2
    assert method(np.complex64, m=3) == 3 and method(np.complex64) == 3 and method(np.complex64, m=6) == 3 and method(np.complex64, m=7) == 2
    os.environ["NUFFT_SMARCH_SPLIT"] = "0"          # the interleaved complex instantiations: ComplexF64 up to M = 4, ComplexF32 up to M = 3
    try:
        assert method(np.complex128) == 3 and method(np.complex128, m=5) == 2
        assert method(np.complex64, m=3) == 3 and method(np.complex64) == 2
    finally:
        del os.environ["NUFFT_SMARCH_SPLIT"]
    assert method(np.complex128, (35, 64, 40), sigma=1.5) == 1 and method(np.float64, (64, 64)) == 1
    # real plans with ntransforms = 2 / 3: the ring spreads the components one after the other (7.3 against 7.5 ms at C4); an
    # explicit request for the patches still spreads them together
    i3 = nufft.PlanNUFFT(np.float64, (64, 64, 64), ntransforms=3, backend=nufft.ROCBackend(0)).info()
    assert i3.spread_method == 3 and i3.patch_planar == 0 and i3.ring_column[0] > 0
    i3p = nufft.PlanNUFFT(np.float64, (64, 64, 64), ntransforms=3, spread_method="mfma_patches", backend=nufft.ROCBackend(0)).info()
    assert i3p.spread_method == 2 and i3p.patch_planar == 3
    with pytest.raises(ValueError):
        nufft.PlanNUFFT(np.float64, (64, 64), spread_method="marching_ring", backend=nufft.ROCBackend(0))


@pytest.mark.parametrize("kw", [dict(gpu_method="global_memory"),
                                dict(gpu_method="global_memory", sort_points=True, block_size=(8, 8, 8)),
                                dict(gpu_method="shared_memory", block_size=(4, 4, 4)),
                                dict(sort_points=True)], ids=lambda k: ",".join(f"{a}={b}" for a, b in k.items()))
@pytest.mark.parametrize("Z", [np.float64, np.complex64])
def test_scheduling_only_arguments_match_oracle(Z, kw):
    """`gpu_method = :global_memory`, `sort_points = True()` and `block_size` only reschedule the same sums in the
    reference (src/spreading/gpu.jl:168-186, src/blocking/gpu.jl:41-69): the transforms of a plan created with them are
    held to the same oracle bounds (the dims of test/pseudo_gpu.jl:109)."""
    _check_type1_type2(Z, (35, 64, 40), 4, 1.5, O.DIRECT, 1, **kw)


def _check_type1_type2(Z, dims, M, sigma, evalmode, C, expect_engine=None, **kw):
    Np = 2000
    nufft, plan, oplan, xs, vs = _make_case(Z, dims, M, sigma, evalmode, C, Np, seed=42, **kw)
    if expect_engine is not None:
        assert plan.info().spread_method == {"lds_tiles": 1, "mfma_patches": 2, "marching_ring": 3}[expect_engine]
    dev = plan.device
    xd = tuple(torch.from_numpy(x).to(dev) for x in xs)
    vd = tuple(torch.from_numpy(v).to(dev) for v in vs)
    nufft.set_points(plan, xd)
    O.set_points(oplan, xs)

    # type 1
    us = tuple(torch.empty(plan.shape, dtype=plan.eltype, device=dev) for _ in range(C))
    nufft.exec_type1(us if C > 1 else us[0], plan, vd if C > 1 else vd[0])
    vso = _oracle_inputs(oplan, vs)
    ref = O.exec_type1(oplan, vso if C > 1 else vso[0])
    ref = ref if C > 1 else [ref]
    # the reference's own bounds (test/pseudo_gpu.jl:159-171), also where the Float64 oracle stands in for Float32:
    # it locates the points in Float32 (coord_dtype), so only window rounding and summation order differ
    tol = _rtol(Z)
    for c in range(C):
        assert _rel(us[c].cpu().numpy(), ref[c]) < tol

    # type 2 (random spectrum)
    rng = np.random.default_rng(7)
    ws = [(rng.standard_normal(plan.shape) + 1j * rng.standard_normal(plan.shape)).astype(oplan.cdtype) for _ in range(C)]
    ws = [w.astype(np.complex64 if plan_real_dtype(Z) == np.float32 else np.complex128) for w in ws]
    wd = tuple(torch.from_numpy(w).to(dev) for w in ws)
    out = tuple(torch.empty(Np, dtype=plan.Z, device=dev) for _ in range(C))
    nufft.exec_type2(out if C > 1 else out[0], plan, wd if C > 1 else wd[0])
    wso = _oracle_inputs(oplan, ws)
    ref2 = O.exec_type2(oplan, wso if C > 1 else wso[0])
    ref2 = ref2 if C > 1 else [ref2]
    for c in range(C):
        assert _rel(out[c].cpu().numpy(), ref2[c]) < tol


@pytest.mark.parametrize("Z", [np.float64, np.complex64])
def test_spread_and_interp_stages_match_oracle(Z):
    """Stage-level parity of spread_from_points! / interpolate! on the oversampled grid itself."""
    dims, M, sigma, Np = (20, 24, 18), 4, 2.0, 3000
    nufft, plan, oplan, xs, vs = _make_case(Z, dims, M, sigma, O.DIRECT, 1, Np, seed=3)
    dev = plan.device
    xd = tuple(torch.from_numpy(x).to(dev) for x in xs)
    nufft.set_points(plan, xd)
    O.set_points(oplan, xs)
    nufft.spread_from_points(plan, torch.from_numpy(vs[0]).to(dev))
    info = plan.info()
    scale = 2.0 ** sum(info.window_scale_log2[d] for d in range(3))     # exact power-of-two window normalisation
    grid_dev = nufft.oversampled_grid(plan).cpu().numpy()
    wide = np.complex128 if np.iscomplexobj(grid_dev) else np.float64
    grid = (grid_dev.astype(wide) / scale).astype(grid_dev.dtype)       # undo the normalisation (rescale in Float64)
    ref = O.spread(oplan, vs)[0]
    assert _rel(grid, ref) < (1e-12 if np.dtype(Z).itemsize >= 8 and Z != np.complex64 else 1e-5)
    # interpolate from the grid that is now in the plan
    out = torch.empty(Np, dtype=plan.Z, device=dev)
    nufft.interpolate(plan, out)
    # The oracle's windows are un-normalised (peak e^β/2π per dimension) and the grid holds raw spread
    # values here, so a Float32 oracle would overflow: evaluate the expectation in Float64 from the same
    # Float32-rounded inputs.
    o64 = O.OraclePlan(dims, is_real=oplan.is_real, dtype=np.float64, M=M, sigma=sigma, evalmode=O.DIRECT)
    O.set_points(o64, [x.astype(np.float64) for x in xs])
    ref2 = O.interpolate(o64, [grid.astype(wide)])[0]
    got = out.cpu().numpy().astype(wide) / scale / scale
    assert _rel(got, ref2) < (1e-12 if Z == np.float64 else 2e-5)


def test_bin_sort_is_a_permutation_grouped_by_bin():
    nufft = _nufft()
    dims, Np = (40, 36, 50), 20000
    plan = nufft.PlanNUFFT(np.float64, dims, m=4, sigma=2.0, backend=nufft.ROCBackend(0))
    rng = np.random.default_rng(0)
    xs = [(rng.random(Np) * 5 - 2) * O.TWO_PI for _ in dims]
    nufft.set_points(plan, tuple(torch.from_numpy(x).cuda() for x in xs))
    perm, offs = nufft.sort_result(plan)
    assert np.array_equal(np.sort(perm), np.arange(Np))                # bijection
    assert offs[0] == 0 and offs[-1] == Np and np.all(np.diff(offs.astype(np.int64)) >= 0)
    info = plan.info()
    Nover = plan.oversampled_dims
    tile = np.zeros(Np, dtype=np.int64)
    mul = 1
    for d in range(3):
        i, _ = O.point_to_cell(O.to_unit_cell(xs[d]), Nover[d])
        i = np.minimum(i, Nover[d] - 1)
        tile += mul * (i // info.bin_dims[d])
        mul *= info.nbins[d]
    sorted_tiles = tile[perm]
    assert np.all(np.diff(sorted_tiles) >= 0)                           # grouped by tile, tiles ascending
    counts = np.bincount(tile, minlength=len(offs) - 1)
    assert np.array_equal(np.diff(offs.astype(np.int64)), counts)


@pytest.mark.parametrize("Z,dims,M", [(np.float64, (40, 36, 50), 4), (np.complex64, (64, 48, 40), 8), (np.float32, (30, 70, 36), 5),
                                      (np.complex128, (24, 130, 20), 3), (np.complex128, (32, 48, 32), 10)])      # (M = 10: bins of 2 cells)
def test_two_level_slab_sort_reproduces_the_fine_sort(Z, dims, M, monkeypatch):
    """3-D plans without a column-layer sort order their points by fine bins in two levels (binsort.hip, CoarseSort::mode = 2: slabs of
    bin rows with LDS histograms, then every slab inside a workgroup's LDS) — the offsets of the sort with global atomics exactly, the
    same points in every bin, and the transforms against the oracle; slabs fuller than a workgroup's LDS are sorted in two passes over
    global memory (the "dense" set), and a point set whose fullest slab exceeds eight capacities (a cluster) takes the sort with global atomics
    (device flag)."""
    nufft = _nufft()
    Np = 30000
    monkeypatch.setenv("NUFFT_COARSE_SORT", "0")
    monkeypatch.setenv("NUFFT_SLAB_MIN_POINTS", "0")
    _, plan, oplan, xs, vs = _make_case(Z, dims, M, 2.0, O.FAST_APPROXIMATION, 1, Np, seed=77 + M)
    monkeypatch.setenv("NUFFT_SLAB_SORT", "0")
    _, plan0, _, _, _ = _make_case(Z, dims, M, 2.0, O.FAST_APPROXIMATION, 1, Np, seed=77 + M)
    dev = plan.device
    rng = np.random.default_rng(3)
    T = plan_real_dtype(Z)
    cluster = tuple((0.01 * rng.standard_normal(Np) + 1.0).astype(T) for _ in dims)      # nearly all points in one slab
    dense = tuple((0.3 * rng.standard_normal(Np) + 2.0).astype(T) for _ in dims)       # slabs fuller than a workgroup's LDS: level 2 in two passes
    for name, pts in (("uniform", xs), ("dense", dense), ("cluster", cluster)):
        xd = tuple(torch.from_numpy(x).to(dev) for x in pts)
        nufft.set_points(plan, xd)
        nufft.set_points(plan0, xd)
        assert plan.sort_method_used() == ("fine_bins" if name == "cluster" else "slabs"), name
        assert plan0.sort_method_used() == "fine_bins"
        perm, offs = nufft.sort_result(plan)
        perm0, offs0 = nufft.sort_result(plan0)
        assert np.array_equal(np.sort(perm), np.arange(Np))
        assert np.array_equal(offs, offs0), name
        # the same points in every bin: sort each bin's run of the permutation
        binid = np.repeat(np.arange(len(offs) - 1), np.diff(offs.astype(np.int64)))
        assert np.array_equal(perm[np.lexsort((perm, binid))], perm0[np.lexsort((perm0, binid))]), name
        O.set_points(oplan, pts)
        u = torch.empty(plan.shape, dtype=plan.eltype, device=dev)
        nufft.exec_type1(u, plan, torch.from_numpy(vs[0]).to(dev))
        ref = O.exec_type1(oplan, _oracle_inputs(oplan, vs))
        assert _rel(u.cpu().numpy(), ref[0]) < _rtol(Z), (name, "type 1")
        out = torch.empty(Np, dtype=plan.Z, device=dev)
        nufft.exec_type2(out, plan, u)
        ref2 = O.exec_type2(oplan, _oracle_inputs(oplan, [u.cpu().numpy()])[0])
        assert _rel(out.cpu().numpy(), ref2) < _rtol(Z), (name, "type 2")
    # and back: a uniform set after the cluster (the running maximum of the slab loads is cleared by every set_points)
    nufft.set_points(plan, tuple(torch.from_numpy(x).to(dev) for x in xs))
    assert plan.sort_method_used() == "slabs"


def test_edge_points_and_empty_input():
    """prevfloat(2π), prevfloat(π), 0, negative and shifted points (test/near_2pi.jl); Np = 0."""
    nufft = _nufft()
    N, M = 32, 8
    plan = nufft.PlanNUFFT(np.complex128, N, m=M, sigma=1.5, backend=nufft.ROCBackend(0))
    x = np.array([np.nextafter(O.TWO_PI, 0.0), np.nextafter(np.pi, 0.0), 0.0, -0.0, -1e-20, O.TWO_PI,
                  -O.TWO_PI, 3 * O.TWO_PI + 0.1, -7.3], dtype=np.float64)
    v = (np.arange(len(x)) + 1.0) * (4.2 + 3j)
    nufft.set_points(plan, torch.from_numpy(x).cuda())
    u = torch.empty(plan.shape, dtype=torch.complex128, device="cuda")
    nufft.exec_type1(u, plan, torch.from_numpy(v).cuda())
    exact = O.nudft_type1([O.fftfreq_int(N)], [x], v)
    assert _rel(u.cpu().numpy(), exact) < 1e-11                          # test/near_2pi.jl:69
    # empty point set: type-1 gives zeros, type-2 gives an empty vector
    nufft.set_points(plan, torch.empty(0, dtype=torch.float64, device="cuda"))
    nufft.exec_type1(u, plan, torch.empty(0, dtype=torch.complex128, device="cuda"))
    assert float(u.abs().max()) == 0.0
    out = torch.empty(0, dtype=torch.complex128, device="cuda")
    nufft.exec_type2(out, plan, u)
    assert out.numel() == 0


def test_clustered_points_single_tile():
    """All points inside one tile (stress for the LDS atomics and the flush)."""
    Z, dims, M = np.float64, (32, 32, 32), 4
    nufft, plan, oplan, xs, vs = _make_case(Z, dims, M, 2.0, O.FAST_APPROXIMATION, 1, 5000, seed=11)
    xs = [(0.05 + 0.02 * np.random.default_rng(d).random(5000)).astype(np.float64) for d in range(3)]
    dev = plan.device
    nufft.set_points(plan, tuple(torch.from_numpy(x).to(dev) for x in xs))
    O.set_points(oplan, xs)
    u = torch.empty(plan.shape, dtype=plan.eltype, device=dev)
    nufft.exec_type1(u, plan, torch.from_numpy(vs[0]).to(dev))
    assert _rel(u.cpu().numpy(), O.exec_type1(oplan, vs[0])) < 1e-7


def test_errors_mirror_the_reference():
    nufft = _nufft()
    with pytest.raises(ValueError):                 # test/errors.jl:5-10 (Ñ < 2M)
        nufft.PlanNUFFT(np.float64, 4, m=8, sigma=1.25, backend=nufft.ROCBackend(0))
    plan = nufft.PlanNUFFT(np.float64, (16, 16), backend=nufft.ROCBackend(0))
    x = torch.zeros(10, dtype=torch.float64, device="cuda")
    with pytest.raises(ValueError):                 # exec before set_points
        nufft.exec_type1(torch.empty(plan.shape, dtype=torch.complex128, device="cuda"), plan, x)
    with pytest.raises(ValueError):                 # wrong precision of the points, src/set_points.jl:35
        nufft.set_points(plan, (x.float(), x.float()))
    with pytest.raises(nufft.DimensionMismatch):    # different lengths, src/blocking/gpu.jl:86
        nufft.set_points(plan, (x, x[:5].contiguous()))
    nufft.set_points(plan, (x, x))
    with pytest.raises(ValueError):                 # complex64 output for a Float64 plan, src/NonuniformFFTs.jl:154
        nufft.exec_type1(torch.empty(plan.shape, dtype=torch.complex64, device="cuda"), plan, x)
    with pytest.raises(nufft.DimensionMismatch):    # wrong uniform shape, :92-103
        nufft.exec_type1(torch.empty((16, 16), dtype=torch.complex128, device="cuda"), plan, x)
    with pytest.raises(nufft.DimensionMismatch):    # wrong number of values, :105-114
        nufft.exec_type1(torch.empty(plan.shape, dtype=torch.complex128, device="cuda"), plan, x[:3].contiguous())




@pytest.mark.parametrize("dist", ["uniform", "cluster"])
def test_automatic_engine_choice_per_point_set(dist):
    """A plan whose engine is the MFMA patches (automatic for complex data) decides at set_points, on the device, which
    engine spreads THIS point set: the patches for well-spread points, the LDS tiles (which can share a heavy tile
    between workgroups) when a few patch tasks would hold most of the points.  Either way exactly one of the two
    kernels does the work and the result equals the explicitly chosen LDS-tile engine; alternating point sets on one
    plan must switch back and forth."""
    nufft = _nufft()
    n, Np = 64, 400_000
    g = torch.Generator(device="cuda").manual_seed(11)
    sets = {
        "uniform": tuple(torch.rand(Np, dtype=torch.float64, device="cuda", generator=g) * (2 * np.pi) for _ in range(3)),
        "cluster": tuple(torch.randn(Np, dtype=torch.float64, device="cuda", generator=g) * 0.05 + np.pi for _ in range(3)),
    }
    v = torch.randn(Np, dtype=torch.complex128, device="cuda", generator=g)
    auto = nufft.PlanNUFFT(torch.complex128, (n, n, n), m=7, backend=nufft.ROCBackend(0))       # (M = 7: the patches are the automatic choice)
    ref = nufft.PlanNUFFT(torch.complex128, (n, n, n), m=7, spread_method="lds_tiles", backend=nufft.ROCBackend(0))
    assert auto.info().spread_method == 2 and ref.info().spread_method == 1
    for name in (dist, "uniform" if dist == "cluster" else "cluster", dist):
        nufft.set_points(auto, sets[name])
        nufft.set_points(ref, sets[name])
        # the decision itself (nufft_spread_engine_used reads the device flag back): 128^3 oversampled = 512 patch tasks,
        # fewer than the wave slots of the device — the rule compares the heaviest task with twice the mean task there
        assert auto.spread_engine_used() == ("mfma_patches" if name == "uniform" else "lds_tiles"), name
        assert ref.spread_engine_used() == "lds_tiles"
        ua = torch.empty(auto.shape, dtype=torch.complex128, device="cuda")
        ur = torch.empty_like(ua)
        nufft.exec_type1(ua, auto, v)
        nufft.exec_type1(ur, ref, v)
        assert float((ua - ur).norm() / ur.norm()) < 1e-12, name


def test_automatic_engine_choice_on_a_grid_with_more_tasks_than_wave_slots():
    """The same decision where the patch tasks outnumber the resident waves (oversampled 512 x 256 x 256: 8192 tasks): the
    rule is then `heaviest task <= np / wave slots`."""
    nufft = _nufft()
    Np = 2_000_000
    g = torch.Generator(device="cuda").manual_seed(12)
    auto = nufft.PlanNUFFT(torch.complex64, (256, 128, 128), m=7, backend=nufft.ROCBackend(0))      # (M = 7: the patches are the automatic choice)
    assert auto.info().spread_method == 2
    for name in ("uniform", "cluster", "uniform"):
        if name == "uniform":
            xs = tuple(torch.rand(Np, dtype=torch.float32, device="cuda", generator=g) * (2 * np.pi) for _ in range(3))
        else:
            xs = tuple(torch.randn(Np, dtype=torch.float32, device="cuda", generator=g) * 0.05 + np.pi for _ in range(3))
        nufft.set_points(auto, xs)
        assert auto.spread_engine_used() == ("mfma_patches" if name == "uniform" else "lds_tiles"), name


