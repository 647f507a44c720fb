"""Seeded random differential test: HIP path vs oracle over randomly drawn plan configurations (dimensions,
sizes incl. primes and sizes smaller than a tile, M, sigma, kernel, evaluation mode, element type, ntransforms,
fftshift, point distribution).  Complements the hand-picked matrix of tests/test_gpu_parity.py, which mirrors
test/pseudo_gpu.jl."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import nufft_oracle as O  # noqa: E402

KERNELS = [(O.KERNEL_BKB, "BackwardsKaiserBesselKernel"), (O.KERNEL_KB, "KaiserBesselKernel"),
           (O.KERNEL_GAUSSIAN, "GaussianKernel"), (O.KERNEL_BSPLINE, "BSplineKernel")]


def _draw(rng):
    D = int(rng.integers(1, 4))
    M = int(rng.integers(2, 11))
    sigma = float(rng.choice([1.25, 1.5, 2.0, 2.0]))
    hi = {1: 400, 2: 70, 3: 28}[D]
    dims = tuple(int(rng.integers(max(2, int(np.ceil(2 * M / sigma))), hi)) for _ in range(D))
    Z = [np.float64, np.complex128, np.float32, np.complex64][int(rng.integers(0, 4))]
    kid, kname = KERNELS[int(rng.integers(0, 4))]
    if kid in (O.KERNEL_GAUSSIAN, O.KERNEL_BSPLINE):
        sigma = 2.0       # as in the reference's tests (test/accuracy.jl:51-76): at small sigma 1 / phi_hat spans 5 decades
    mode = int(rng.integers(0, 2))
    C = int(rng.choice([1, 1, 2, 3]))
    fftshift = bool(rng.integers(0, 2))
    dist = ["uniform", "cluster", "edges"][int(rng.integers(0, 3))]
    return D, M, sigma, dims, Z, kid, kname, mode, C, fftshift, dist


@pytest.mark.parametrize("seed", range(24))
def test_random_configuration(seed):
    from nufft_pkg import nufft
    rng = np.random.default_rng(1000 + seed)
    D, M, sigma, dims, Z, kid, kname, mode, C, fftshift, dist = _draw(rng)
    Zt = np.dtype(Z)
    is_real = Zt.kind == "f"
    T = np.float32 if Zt in (np.dtype(np.float32), np.dtype(np.complex64)) else np.float64
    # Float32 plans are checked against the Float32 oracle (same fold arithmetic: a point at a multiple of 2 pi may
    # land on either side of the periodic boundary in Float32, which matters at the window-truncation level,
    # 1e-2 at M = 2) unless its un-normalised windows overflow (D * M >= 18, conservatively): then against the Float64 oracle.
    big_window = kid in (O.KERNEL_BKB, O.KERNEL_KB)
    To = np.float64 if (T == np.float64 or (big_window and D * M >= 18)) else np.float32
    try:
        # (Float64 oracle for a Float32 plan: the points are located in Float32 exactly as the plan does, coord_dtype)
        oplan64 = O.OraclePlan(dims, is_real=is_real, dtype=To, coord_dtype=(T if To != T else None), M=M, sigma=sigma, evalmode=mode,
                               ntransforms=C, kernel=kid, fftshift=fftshift)
    except ValueError:
        pytest.skip("oversampled size below 2M")
    Np = int(rng.integers(1, 3000))
    if dist == "uniform":
        xs = [(rng.random(Np) * 3 - 1) * O.TWO_PI for _ in dims]
    elif dist == "cluster":
        xs = [rng.standard_normal(Np) * 0.05 + rng.random() * O.TWO_PI for _ in dims]
    else:
        edge = np.array([0.0, np.nextafter(O.TWO_PI, 0), O.TWO_PI, -1e-300, np.pi, 3 * O.TWO_PI])
        xs = [rng.choice(edge, Np) for _ in dims]
    xs = [x.astype(T) for x in xs]
    vs = [(rng.standard_normal(Np) if is_real else rng.standard_normal(Np) + 1j * rng.standard_normal(Np)).astype(Zt)
          for _ in range(C)]
    plan = nufft.PlanNUFFT(Zt, dims, m=M, sigma=sigma, ntransforms=C, kernel=getattr(nufft, kname)(), fftshift=fftshift,
                           kernel_evalmode=nufft.Direct() if mode == O.DIRECT else nufft.FastApproximation(),
                           backend=nufft.ROCBackend(0))
    assert plan.oversampled_dims == oplan64.Nover
    O.set_points(oplan64, [x.astype(To) for x in xs] if To == T else xs)
    dev = plan.device
    nufft.set_points(plan, tuple(torch.from_numpy(x).to(dev) for x in xs))
    us = tuple(torch.empty(plan.shape, dtype=plan.eltype, device=dev) for _ in range(C))
    vd = tuple(torch.from_numpy(v).to(dev) for v in vs)
    nufft.exec_type1(us if C > 1 else us[0], plan, vd if C > 1 else vd[0])
    cT = np.complex128 if To == np.float64 else np.complex64
    v64 = [v.astype(cT if not is_real else To) for v in vs]
    ref = O.exec_type1(oplan64, v64 if C > 1 else v64[0])
    ref = ref if C > 1 else [ref]
    # the reference's own bounds (test/pseudo_gpu.jl:159-171): 1e-7 Float64, 1e-5 Float32 — measured over all seeds (NUFFT_TEST_ERRLOG):
    # Float32 at most 3.8e-6 (the wide-window case compares with the Float64 oracle that locates the points in Float32)
    tol = 1e-7 if T == np.float64 else 1e-5
    for c in range(C):
        denom = np.linalg.norm(ref[c].ravel())
        err = np.linalg.norm((us[c].cpu().numpy() - ref[c]).ravel())
        _log_error("small", seed, T, To, err / max(denom, 1e-30), (dims, M, sigma, kname, mode, str(Zt), dist, "type 1"))
        assert err <= tol * max(denom, 1e-30), (dims, M, sigma, kname, mode, Zt, dist)
    ws = [(rng.standard_normal(plan.shape) + 1j * rng.standard_normal(plan.shape)).astype(np.complex64 if T == np.float32 else np.complex128)
          for _ in range(C)]
    out = tuple(torch.empty(Np, dtype=plan.Z, device=dev) for _ in range(C))
    wd = tuple(torch.from_numpy(w).to(dev) for w in ws)
    nufft.exec_type2(out if C > 1 else out[0], plan, wd if C > 1 else wd[0])
    w64 = [w.astype(cT) for w in ws]
    ref2 = O.exec_type2(oplan64, w64 if C > 1 else w64[0])
    ref2 = ref2 if C > 1 else [ref2]
    for c in range(C):
        denom = np.linalg.norm(np.asarray(ref2[c]).ravel())
        err = np.linalg.norm((out[c].cpu().numpy() - ref2[c]).ravel())
        _log_error("small", seed, T, To, err / max(denom, 1e-30), (dims, M, sigma, kname, mode, str(Zt), dist, "type 2"))
        assert err <= tol * max(denom, 1e-30), (dims, M, sigma, kname, mode, Zt, dist)


def _log_error(kind, seed, T, To, err, what):
    """NUFFT_TEST_ERRLOG=<file>: append the measured relative error of every Float32 comparison (soak / tolerance audits)."""
    path = os.environ.get("NUFFT_TEST_ERRLOG")
    if path and T == np.float32:
        with open(path, "a") as fh:
            fh.write(f"{kind} seed={seed} oracle={'f32' if To == np.float32 else 'f64'} err={err:.3e} {what}\n")


def _draw_large(rng):
    M = int(rng.integers(2, 11))
    sigma = float(rng.choice([1.25, 1.5, 2.0, 2.0]))
    dims = tuple(int(rng.integers(28, 72)) for _ in range(3))
    Z = [np.float64, np.complex128, np.float32, np.complex64][int(rng.integers(0, 4))]
    mode = int(rng.integers(0, 2))
    C = int(rng.choice([1, 1, 2, 3]))
    dist = ["uniform", "uniform", "cluster"][int(rng.integers(0, 3))]
    engine = ["auto", "auto", "lds_tiles", "mfma_patches", "marching_ring"][int(rng.integers(0, 5))]
    if np.dtype(Z) in (np.dtype(np.float32), np.dtype(np.complex64)):
        # Float32 at small sigma is ill-conditioned whatever the implementation: the deconvolution spans 4-5 decades and amplifies
        # Float32 round-off of windows and sums to 1e-3 (measured at sigma = 1.25, M = 6, clustered points: Float32 oracle vs
        # Float64-accumulating oracle 2.6e-3, HIP path vs the latter 7.7e-4) — nothing to compare at 1e-5 there
        sigma = 2.0
    return M, sigma, dims, Z, mode, C, dist, engine


@pytest.mark.parametrize("seed", range(56 + int(os.environ.get("NUFFT_TEST_EXTRA_SEEDS", "0"))))      # (soak runs: more seeds)
def test_random_configuration_large_3d(seed, monkeypatch):
    """The same differential test on 3-D grids large enough for the engines that need room — register patches (Float64 and
    Float32 accumulators, planar components), the z-marching spreading window and interpolation ring — with the default window, both evaluation
    modes, all element types, ntransforms 1..3, uniform and clustered points (clustered sets switch both stages back to the
    LDS-tile kernels on the device), and the spreading engine automatic or forced (an engine the plan cannot serve must be
    refused, nothing silent)."""
    from nufft_pkg import nufft
    rng = np.random.default_rng(5000 + seed)
    M, sigma, dims, Z, mode, C, dist, engine = _draw_large(rng)
    if seed >= 32:
        # seeds 32..55: point sets far from uniform with the interpolation ring forced (and the patches wherever the draw
        # forces them) — the tasks of equal point count that set_points cuts per point set (balance.hip)
        dist = ["randn", "slab", "columns"][seed % 3]
        monkeypatch.setenv("NUFFT_INTERP_MARCH", "2")
    Zt = np.dtype(Z)
    is_real = Zt.kind == "f"
    T = np.float32 if Zt in (np.dtype(np.float32), np.dtype(np.complex64)) else np.float64
    wide = T == np.float32 and 3 * M >= 18             # un-normalised Float32 windows overflow: Float64 oracle, Float32 cell arithmetic
    oplan = O.OraclePlan(dims, is_real=is_real, dtype=np.float64 if wide else T, coord_dtype=T if wide else None, M=M, sigma=sigma,
                         evalmode=mode, ntransforms=C)
    Np = int(rng.integers(500, 6000))
    if dist == "uniform":
        xs = [(rng.random(Np) * 3 - 1) * O.TWO_PI for _ in dims]
    elif dist == "randn":                              # the reference's benchmark distribution (folded by the library)
        xs = [rng.standard_normal(Np) for _ in dims]
    elif dist == "slab":                               # a few bin layers along dimension 3 hold every point
        xs = [rng.random(Np) * O.TWO_PI, rng.random(Np) * O.TWO_PI, rng.standard_normal(Np) * 0.05 + rng.random() * O.TWO_PI]
    elif dist == "columns":                            # a few columns of the grid hold every point
        xs = [rng.standard_normal(Np) * 0.08 + rng.random() * O.TWO_PI, rng.standard_normal(Np) * 0.08 + rng.random() * O.TWO_PI,
              rng.random(Np) * O.TWO_PI]
    else:
        xs = [rng.standard_normal(Np) * 0.1 + rng.random() * O.TWO_PI for _ in dims]
    xs = [x.astype(T) for x in xs]
    vs = [(rng.standard_normal(Np) if is_real else rng.standard_normal(Np) + 1j * rng.standard_normal(Np)).astype(Zt)
          for _ in range(C)]
    try:
        plan = nufft.PlanNUFFT(Zt, dims, m=M, sigma=sigma, ntransforms=C, spread_method=engine,
                               kernel_evalmode=nufft.Direct() if mode == O.DIRECT else nufft.FastApproximation(),
                               backend=nufft.ROCBackend(0))
    except ValueError as exc:
        assert (engine == "mfma_patches" and "MFMA patches" in str(exc)) or (engine == "marching_ring" and "marching ring" in str(exc))
        return
    assert plan.oversampled_dims == oplan.Nover
    O.set_points(oplan, xs)
    dev = plan.device
    nufft.set_points(plan, tuple(torch.from_numpy(x).to(dev) for x in xs))
    if engine != "auto":
        assert plan.spread_engine_used() == engine
    us = tuple(torch.empty(plan.shape, dtype=plan.eltype, device=dev) for _ in range(C))
    vd = tuple(torch.from_numpy(v).to(dev) for v in vs)
    nufft.exec_type1(us if C > 1 else us[0], plan, vd if C > 1 else vd[0])
    wide_c = np.complex128 if (wide or T == np.float64) else np.complex64
    vo = [v.astype(wide_c if not is_real else (np.float64 if wide else T)) for v in vs]
    ref = O.exec_type1(oplan, vo if C > 1 else vo[0])
    ref = ref if C > 1 else [ref]
    tol = 1e-7 if T == np.float64 else 1e-5             # the reference's bounds (test/pseudo_gpu.jl:159-171); Float32 measured <= 3.2e-6 over all seeds
    # Float64 at the ill-conditioned corner of the draw (M = 10 at sigma = 1.25: the deconvolution spans prod_d max / min |phi_hat_d|
    # = 6.8e11, so two summation orders of the same sums differ by 1e-7 — measured: the LDS tiles against themselves 1.1e-7, every
    # engine 3-4e-7 against the oracle, soak seed 375): the bound follows the conditioning there, 1e-7 everywhere else (cond <= 5e10)
    if T == np.float64:
        cond = float(np.prod([np.abs(ph).max() / np.abs(ph).min() for ph in oplan.phihat]))
        tol = max(tol, 2e-18 * cond)
    for c in range(C):
        err = np.linalg.norm(us[c].cpu().numpy().astype(np.complex128) - ref[c]) / np.linalg.norm(ref[c])
        _log_error("large", seed, T, np.float64 if wide else T, err, (dims, M, sigma, mode, str(Zt), C, dist, engine, "type 1"))
        assert err < tol, (seed, M, sigma, dims, Z, mode, C, dist, engine, err)
    ws = [(rng.standard_normal(plan.shape) + 1j * rng.standard_normal(plan.shape)).astype(np.complex64 if T == np.float32 else np.complex128)
          for _ in range(C)]
    wd = tuple(torch.from_numpy(w).to(dev) for w in ws)
    out = tuple(torch.empty(Np, dtype=plan.Z, device=dev) for _ in range(C))
    nufft.exec_type2(out if C > 1 else out[0], plan, wd if C > 1 else wd[0])
    wo = [w.astype(wide_c) for w in ws]
    ref2 = O.exec_type2(oplan, wo if C > 1 else wo[0])
    ref2 = ref2 if C > 1 else [ref2]
    for c in range(C):
        err = np.linalg.norm(out[c].cpu().numpy() - ref2[c]) / np.linalg.norm(ref2[c])
        _log_error("large", seed, T, np.float64 if wide else T, err, (dims, M, sigma, mode, str(Zt), C, dist, engine, "type 2"))
        assert err < tol, (seed, M, sigma, dims, Z, mode, C, dist, engine, "type 2", err)
