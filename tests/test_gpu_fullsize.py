"""Full-size parity at BASELINE config C2 (256^3, Np = 1e7, Float64, m = 4, sigma = 2), where the CPU
oracle is too slow: size-independent properties and exact spot checks computed on the GPU with plain
torch arithmetic (O(Np) per mode, O(N^3) per point)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

N, NP, M, SIGMA = 256, 10_000_000, 4, 2.0
CEIL = 6 * 10.0 ** (-1.9 * M)        # test/accuracy.jl:33-35 (1.5e-7 at m = 4, sigma = 2)


@pytest.fixture(scope="module")
def case():
    from nufft_pkg import nufft
    plan = nufft.PlanNUFFT(torch.float64, (N, N, N), m=M, sigma=SIGMA, backend=nufft.ROCBackend(0))
    g = torch.Generator(device="cuda").manual_seed(1234)
    xs = tuple(torch.rand(NP, dtype=torch.float64, device="cuda", generator=g) * (2 * np.pi) for _ in range(3))
    v = torch.randn(NP, dtype=torch.float64, device="cuda", generator=g)
    nufft.set_points(plan, xs)
    u = torch.empty(plan.shape, dtype=torch.complex128, device="cuda")
    nufft.exec_type1(u, plan, v)
    return nufft, plan, xs, v, u


def _wavenumbers(plan):
    k1 = torch.arange(N // 2 + 1, dtype=torch.float64, device="cuda")
    k = torch.arange(N, dtype=torch.float64, device="cuda")
    k = torch.where(k >= (N + 1) // 2, k - N, k)
    return k1, k, k


def test_type1_spot_modes_against_exact_sum(case):
    nufft, plan, xs, v, u = case
    k1, k2, k3 = _wavenumbers(plan)
    rng = np.random.default_rng(0)
    num = den = 0.0
    for _ in range(24):
        i1, i2, i3 = int(rng.integers(0, N // 2 + 1)), int(rng.integers(0, N)), int(rng.integers(0, N))
        phase = k1[i1] * xs[0] + k2[i2] * xs[1] + k3[i3] * xs[2]
        exact = torch.complex((v * torch.cos(phase)).sum(), -(v * torch.sin(phase)).sum())
        got = u[i3, i2, i1]
        num += float((got - exact).abs() ** 2)
        den += float(exact.abs() ** 2)
    assert np.sqrt(num / den) < 2 * CEIL


def test_type2_spot_points_against_exact_sum(case):
    nufft, plan, xs, v, u = case
    g = torch.Generator(device="cuda").manual_seed(7)
    w = torch.complex(torch.randn(plan.shape, dtype=torch.float64, device="cuda", generator=g),
                      torch.randn(plan.shape, dtype=torch.float64, device="cuda", generator=g))
    out = torch.empty(NP, dtype=torch.float64, device="cuda")
    nufft.exec_type2(out, plan, w)
    k1, k2, k3 = _wavenumbers(plan)
    h = torch.full((N // 2 + 1,), 2.0, dtype=torch.float64, device="cuda")
    h[0] = 1.0                                    # Hermitian weights, test/accuracy.jl:184-186
    num = den = 0.0
    for j in np.random.default_rng(1).integers(0, NP, 16):
        e1 = torch.polar(h, k1 * xs[0][j])
        e2 = torch.polar(torch.ones_like(k2), k2 * xs[1][j])
        e3 = torch.polar(torch.ones_like(k3), k3 * xs[2][j])
        exact = torch.einsum("cba,c,b,a->", w, e3, e2, e1).real
        num += float((out[j] - exact) ** 2)
        den += float(exact ** 2)
    assert np.sqrt(num / den) < 2 * CEIL


def test_linearity_and_adjointness(case):
    nufft, plan, xs, v, u = case
    g = torch.Generator(device="cuda").manual_seed(99)
    v2 = torch.randn(NP, dtype=torch.float64, device="cuda", generator=g)
    u2 = torch.empty_like(u)
    nufft.exec_type1(u2, plan, v2)
    u3 = torch.empty_like(u)
    nufft.exec_type1(u3, plan, 0.75 * v - 2.5 * v2)
    lin = (u3 - (0.75 * u - 2.5 * u2)).norm() / u3.norm()
    assert float(lin) < 1e-12                     # atomics reorder the sums: round-off level, not bitwise
    # adjointness: sum_j v_j (T2 w)_j == Re sum_k h(k1) w_k conj((T1 v)_k)
    w = torch.complex(torch.randn(plan.shape, dtype=torch.float64, device="cuda", generator=g),
                      torch.randn(plan.shape, dtype=torch.float64, device="cuda", generator=g))
    out = torch.empty(NP, dtype=torch.float64, device="cuda")
    nufft.exec_type2(out, plan, w)
    h = torch.full((N // 2 + 1,), 2.0, dtype=torch.float64, device="cuda")
    h[0] = 1.0
    lhs = float((v * out).sum())
    rhs = float((h * (w * u.conj()).real).sum())
    scale = float(v.norm() * out.norm())
    assert abs(lhs - rhs) / scale < 1e-6


def test_repeatability_and_reuse(case):
    """Same inputs twice: identical up to the summation order of the atomics; set_points with a smaller
    point set afterwards reuses the plan's buffers correctly."""
    nufft, plan, xs, v, u = case
    ub = torch.empty_like(u)
    nufft.exec_type1(ub, plan, v)
    assert float((ub - u).norm() / u.norm()) < 1e-13
    n = 1000
    xs_small = tuple(x[:n].contiguous() for x in xs)
    nufft.set_points(plan, xs_small)
    us = torch.empty_like(u)
    nufft.exec_type1(us, plan, v[:n].contiguous())
    k1, k2, k3 = _wavenumbers(plan)
    phase = k1[5] * xs_small[0] + k2[250] * xs_small[1] + k3[3] * xs_small[2]
    exact = torch.complex((v[:n] * torch.cos(phase)).sum(), -(v[:n] * torch.sin(phase)).sum())
    assert float((us[3, 250, 5] - exact).abs() / exact.abs()) < 1e-5
    nufft.set_points(plan, xs)                    # restore for other tests


@pytest.mark.parametrize("dist", ["randn", "cluster"])
def test_nonuniform_distributions_spot_check_and_balance(dist):
    """Non-uniform point sets (the reference's benchmark draws folded N(0, 1) coordinates,
    benchmark/CPU+AMDGPU/run_benchmarks.jl:57-66; "cluster": N(pi, 0.1^2), almost every point in a handful of
    tiles): heavy tiles are shared by several workgroups (balance.hip).  Exact spot checks of type-1 modes, and
    type 1 / type 2 against a plan that runs without load balancing (NUFFT_BALANCE=0)."""
    import os
    from nufft_pkg import nufft
    Np = 4_000_000
    g = torch.Generator(device="cuda").manual_seed(77)
    if dist == "randn":
        xs = tuple(torch.randn(Np, dtype=torch.float64, device="cuda", generator=g) for _ in range(3))
    else:
        xs = tuple(torch.randn(Np, dtype=torch.float64, device="cuda", generator=g) * 0.1 + np.pi for _ in range(3))
    v = torch.randn(Np, dtype=torch.float64, device="cuda", generator=g)
    plan = nufft.PlanNUFFT(torch.float64, (N, N, N), m=M, sigma=SIGMA, backend=nufft.ROCBackend(0))
    nufft.set_points(plan, xs)
    u = torch.empty(plan.shape, dtype=torch.complex128, device="cuda")
    nufft.exec_type1(u, plan, v)
    k1, k2, k3 = _wavenumbers(plan)
    rng = np.random.default_rng(5)
    num = den = 0.0
    for _ in range(16):
        i1, i2, i3 = int(rng.integers(0, N // 2 + 1)), int(rng.integers(0, N)), int(rng.integers(0, N))
        phase = k1[i1] * xs[0] + k2[i2] * xs[1] + k3[i3] * xs[2]
        exact = torch.complex((v * torch.cos(phase)).sum(), -(v * torch.sin(phase)).sum())
        num += float((u[i3, i2, i1] - exact).abs() ** 2)
        den += float(exact.abs() ** 2)
    assert np.sqrt(num / den) < 2 * CEIL
    out = torch.empty(Np, dtype=torch.float64, device="cuda")
    nufft.exec_type2(out, plan, u)
    os.environ["NUFFT_BALANCE"] = "0"
    try:
        plain = nufft.PlanNUFFT(torch.float64, (N, N, N), m=M, sigma=SIGMA, backend=nufft.ROCBackend(0))
    finally:
        del os.environ["NUFFT_BALANCE"]
    nufft.set_points(plain, xs)
    u0 = torch.empty_like(u)
    nufft.exec_type1(u0, plain, v)
    assert float((u - u0).norm() / u0.norm()) < 1e-12          # same sums, different order of the atomics
    out0 = torch.empty_like(out)
    nufft.exec_type2(out0, plain, u)
    assert float((out - out0).norm() / out0.norm()) < 1e-13


@pytest.mark.parametrize("mode", ["FastApproximation", "Direct"])
def test_c2_full_grid_rel_l2_against_c_oracle(mode):
    """Full rel-L2 over all 129 x 256 x 256 output modes of the C2 transform at its stated size (256^3, sigma = 2, m = 4,
    Float64, Np = 1e7) against the C restatement of the reference's blocked CPU
    algorithm (oracle/nufft_oracle.c + pocketfft), and of type 2 over all points — with the polynomial window and
    with Direct(), the ROC default (ext/NonuniformFFTsAMDGPUExt.jl:56) and the window of bench.py's headline value."""
    from oracle import c_oracle as CO, nufft_oracle as O
    from nufft_pkg import nufft
    if not CO.available():
        pytest.skip("oracle/libnufft_oracle.so not built")
    Np = NP
    rng = np.random.default_rng(2024)
    xs = [rng.random(Np) * O.TWO_PI for _ in range(3)]
    v = rng.standard_normal(Np)
    oplan = O.OraclePlan((N, N, N), is_real=True, M=M, sigma=SIGMA, evalmode=O.FAST_APPROXIMATION if mode == "FastApproximation" else O.DIRECT)
    O.set_points(oplan, xs)
    ref = CO.exec_type1(oplan, v)
    plan = nufft.PlanNUFFT(torch.float64, (N, N, N), m=M, sigma=SIGMA, backend=nufft.ROCBackend(0),
                           kernel_evalmode=getattr(nufft, mode)())
    nufft.set_points(plan, tuple(torch.from_numpy(x).cuda() for x in xs))
    u = torch.empty(plan.shape, dtype=torch.complex128, device="cuda")
    nufft.exec_type1(u, plan, torch.from_numpy(v).cuda())
    got = u.cpu().numpy()
    e1 = np.linalg.norm((got - ref).ravel()) / np.linalg.norm(ref.ravel())
    out = torch.empty(Np, dtype=torch.float64, device="cuda")
    nufft.exec_type2(out, plan, u)
    ref2 = CO.exec_type2(oplan, got)
    e2 = np.linalg.norm(out.cpu().numpy() - ref2) / np.linalg.norm(ref2)
    print(f"C2 full size, {mode}: type 1 rel-L2 vs C oracle {e1:.2e}, type 2 {e2:.2e}; engines {plan.spread_engine_used()} / "
          f"{plan.interp_engine_used()}, column-layer sort {plan.sort_columns_used()}")
    assert e1 < 1e-11 and e2 < 1e-11, (e1, e2)


def test_config_c4_ntransforms3_spot_check():
    """BASELINE config C4: C2 with ntransforms = 3 (three value vectors spread / interpolated simultaneously,
    one set of points).  Exact spot checks per component; component c must equal a single transform of v_c."""
    from nufft_pkg import nufft
    Np, C = NP, 3                      # the stated size: Np = 1e7
    plan = nufft.PlanNUFFT(torch.float64, (N, N, N), m=M, sigma=SIGMA, ntransforms=C, backend=nufft.ROCBackend(0),
                           kernel_evalmode=nufft.FastApproximation())
    g = torch.Generator(device="cuda").manual_seed(31)
    xs = tuple(torch.rand(Np, dtype=torch.float64, device="cuda", generator=g) * (2 * np.pi) for _ in range(3))
    vs = tuple(torch.randn(Np, dtype=torch.float64, device="cuda", generator=g) for _ in range(C))
    nufft.set_points(plan, xs)
    us = tuple(torch.empty(plan.shape, dtype=torch.complex128, device="cuda") for _ in range(C))
    nufft.exec_type1(us, plan, vs)
    k1, k2, k3 = _wavenumbers(plan)
    rng = np.random.default_rng(9)
    for c in range(C):
        num = den = 0.0
        for _ in range(8):
            i1, i2, i3 = int(rng.integers(0, N // 2 + 1)), int(rng.integers(0, N)), int(rng.integers(0, N))
            phase = k1[i1] * xs[0] + k2[i2] * xs[1] + k3[i3] * xs[2]
            exact = torch.complex((vs[c] * torch.cos(phase)).sum(), -(vs[c] * torch.sin(phase)).sum())
            num += float((us[c][i3, i2, i1] - exact).abs() ** 2)
            den += float(exact.abs() ** 2)
        assert np.sqrt(num / den) < 2 * CEIL
    single = nufft.PlanNUFFT(torch.float64, (N, N, N), m=M, sigma=SIGMA, backend=nufft.ROCBackend(0),
                             kernel_evalmode=nufft.FastApproximation())
    nufft.set_points(single, xs)
    u1 = torch.empty(plan.shape, dtype=torch.complex128, device="cuda")
    nufft.exec_type1(u1, single, vs[1])
    assert float((u1 - us[1]).norm() / u1.norm()) < 1e-12
    outs = tuple(torch.empty(Np, dtype=torch.float64, device="cuda") for _ in range(C))
    nufft.exec_type2(outs, plan, us)
    # type 2 of every component against the exact sum over all 129 x 256 x 256 modes on spot points (Hermitian weights of the
    # half spectrum, test/accuracy.jl:184-186) — the three components carry different spectra, so a mixed-up component fails
    h = torch.full((N // 2 + 1,), 2.0, dtype=torch.float64, device="cuda")
    h[0] = 1.0
    pts = np.random.default_rng(17).integers(0, Np, 6)
    for c in range(C):
        num = den = 0.0
        for j in pts:
            e1 = torch.polar(h, k1 * xs[0][j])
            e2 = torch.polar(torch.ones_like(k2), k2 * xs[1][j])
            e3 = torch.polar(torch.ones_like(k3), k3 * xs[2][j])
            exact = torch.einsum("cba,c,b,a->", us[c], e3, e2, e1).real
            num += float((outs[c][j] - exact) ** 2)
            den += float(exact ** 2)
        assert np.sqrt(num / den) < 2 * CEIL, c
    o1 = torch.empty(Np, dtype=torch.float64, device="cuda")
    nufft.exec_type2(o1, single, us[1])
    assert float((o1 - outs[1]).norm() / o1.norm()) < 1e-12


def test_c4_full_grid_rel_l2_of_every_component_against_c_oracle():
    """BASELINE config C4 at its stated size (256^3, Np = 1e7, Float64, m = 4, ntransforms = 3) with Direct(), the ROC default: full
    rel-L2 over all 129 x 256 x 256 modes of EVERY component against the C oracle's three-component transform of the same inputs, and
    of type 2 over all points of every component (next to the exact spot checks above).  The components
    carry independent values, so a swapped or shared component cannot pass."""
    from oracle import c_oracle as CO, nufft_oracle as O
    from nufft_pkg import nufft
    if not CO.available():
        pytest.skip("oracle/libnufft_oracle.so not built")
    Np, C = NP, 3
    rng = np.random.default_rng(4004)
    xs = [rng.random(Np) * O.TWO_PI for _ in range(3)]
    vs = [rng.standard_normal(Np) for _ in range(C)]
    oplan = O.OraclePlan((N, N, N), is_real=True, M=M, sigma=SIGMA, evalmode=O.DIRECT, ntransforms=C)
    O.set_points(oplan, xs)
    refs = CO.exec_type1(oplan, vs)
    plan = nufft.PlanNUFFT(torch.float64, (N, N, N), m=M, sigma=SIGMA, ntransforms=C, backend=nufft.ROCBackend(0), kernel_evalmode=nufft.Direct())
    nufft.set_points(plan, tuple(torch.from_numpy(x).cuda() for x in xs))
    us = tuple(torch.empty(plan.shape, dtype=torch.complex128, device="cuda") for _ in range(C))
    nufft.exec_type1(us, plan, tuple(torch.from_numpy(v).cuda() for v in vs))
    got = [u.cpu().numpy() for u in us]
    e1 = [float(np.linalg.norm((got[c] - refs[c]).ravel()) / np.linalg.norm(refs[c].ravel())) for c in range(C)]
    # cross terms: component c against the oracle's component c' != c must be far off (independent values)
    assert float(np.linalg.norm((got[0] - refs[1]).ravel()) / np.linalg.norm(refs[1].ravel())) > 0.5
    outs = tuple(torch.empty(Np, dtype=torch.float64, device="cuda") for _ in range(C))
    nufft.exec_type2(outs, plan, us)
    ref2 = CO.exec_type2(oplan, got)
    e2 = [float(np.linalg.norm(outs[c].cpu().numpy() - ref2[c]) / np.linalg.norm(ref2[c])) for c in range(C)]
    print(f"C4 full size, Direct: type 1 rel-L2 per component vs C oracle {e1}, type 2 {e2}; engines {plan.spread_engine_used()} / "
          f"{plan.interp_engine_used()}, column-layer sort {plan.sort_columns_used()}")
    assert max(e1) < 1e-11 and max(e2) < 1e-11, (e1, e2)


def test_config_c3_complexf32_m8_1024_cubed_spot_check():
    """BASELINE config C3 shape: 3-D, Ns = 512^3, ComplexF32, m = 8 (oversampled 1024^3, 8.6 GB grid; LDS pressure).
    at the stated Np = 1e8; exact spot checks of type-1 modes and type-2 points in Float64.
    The reference's un-normalised Float32 window overflows at this (D, M) — see DESIGN.md §2."""
    from nufft_pkg import nufft
    Ns, Np, M8 = 512, 100_000_000, 8
    plan = nufft.PlanNUFFT(torch.complex64, (Ns, Ns, Ns), m=M8, sigma=2.0, backend=nufft.ROCBackend(0),
                           kernel_evalmode=nufft.FastApproximation())
    assert plan.oversampled_dims == (1024, 1024, 1024)
    g = torch.Generator(device="cuda").manual_seed(5)
    xs = tuple(torch.rand(Np, dtype=torch.float32, device="cuda", generator=g) * (2 * np.pi) for _ in range(3))
    v = torch.complex(torch.randn(Np, dtype=torch.float32, device="cuda", generator=g),
                      torch.randn(Np, dtype=torch.float32, device="cuda", generator=g))
    nufft.set_points(plan, xs)
    u = torch.empty(plan.shape, dtype=torch.complex64, device="cuda")
    nufft.exec_type1(u, plan, v)
    assert bool(torch.isfinite(torch.view_as_real(u)).all())
    k = torch.arange(Ns, dtype=torch.float64, device="cuda")
    k = torch.where(k >= (Ns + 1) // 2, k - Ns, k)
    x64 = [x.double() for x in xs]
    v64 = v.to(torch.complex128)
    rng = np.random.default_rng(3)
    num = den = 0.0
    errs = []
    for _ in range(256):
        i1, i2, i3 = (int(rng.integers(0, Ns)) for _ in range(3))
        phase = k[i1] * x64[0] + k[i2] * x64[1] + k[i3] * x64[2]
        exact = (v64 * torch.polar(torch.ones_like(phase), -phase)).sum()
        errs.append(float((u[i3, i2, i1].to(torch.complex128) - exact).abs()))
        num += errs[-1] ** 2
        den += float(exact.abs() ** 2)
    # Float32 data: the bound is the Float32 round-off of coordinates (k x with |k| <= 256 and x in Float32: 256 * 2 pi * 6e-8 = 1e-4 of a
    # radian per point, random in sign) and of sums over 1e8 Float32 terms, not the m = 8 window (1e-14): 256 random modes.  Measured
    # 3.2e-5 for both types (printed; round 5); the bar is 5e-5 — the reference's own Float32 criterion is 1e-5 against its Float32 CPU path, which
    # shares the coordinate rounding, while this is against exact Float64 sums (the dense 128^3 case below holds 1e-5 against the C oracle)
    e1 = np.sqrt(num / den)
    print(f"C3 full size: type 1 rel-L2 over 256 modes vs exact sums {e1:.2e} (largest deviation {max(errs):.3e})")
    assert e1 < 5e-5, (e1, max(errs))
    # type 2 back from a smooth random spectrum: spot-check points
    w = torch.complex(torch.randn(plan.shape, dtype=torch.float32, device="cuda", generator=g),
                      torch.randn(plan.shape, dtype=torch.float32, device="cuda", generator=g))
    out = torch.empty(Np, dtype=torch.complex64, device="cuda")
    nufft.exec_type2(out, plan, w)
    w64 = w.to(torch.complex128)
    num = den = 0.0
    npts2 = 64
    for j in rng.integers(0, Np, npts2):
        e1 = torch.polar(torch.ones_like(k), k * x64[0][j])
        e2 = torch.polar(torch.ones_like(k), k * x64[1][j])
        e3 = torch.polar(torch.ones_like(k), k * x64[2][j])
        exact = torch.einsum("cba,c,b,a->", w64, e3, e2, e1)
        num += float((out[j].to(torch.complex128) - exact).abs() ** 2)
        den += float(exact.abs() ** 2)
    e2 = np.sqrt(num / den)
    print(f"C3 full size: type 2 rel-L2 over {npts2} points vs exact sums {e2:.2e}")
    assert e2 < 5e-5, e2


def test_complexf32_m8_dense_128_cubed_against_c_oracle():
    """The C3 kernels (ComplexF32, m = 8: FP32-matrix-pipe patches, interpolation ring with paired rows) at a density and grid
    where run tables, chunks and K-batches are all at capacity and the host can still check everything: Ns = 128^3
    (oversampled 256^3), Np = 1.6e7 (61 points per bin), full rel-L2 over all modes (type 1) and all points (type 2) against the
    C oracle in Float64 with the points located in Float32 (the reference's un-normalised Float32 window overflows at this
    (D, M): DESIGN.md section 2), at the reference's Float32 bound 1e-5 (test/pseudo_gpu.jl:159-171)."""
    from nufft_pkg import nufft
    from oracle import nufft_oracle as O, c_oracle as CO
    if not CO.available():
        pytest.skip("C oracle not built")
    n, Np, M8 = 128, 16_000_000, 8
    rng = np.random.default_rng(88)
    xs = [(rng.random(Np) * O.TWO_PI).astype(np.float32) for _ in range(3)]
    v = (rng.standard_normal(Np) + 1j * rng.standard_normal(Np)).astype(np.complex64)
    plan = nufft.PlanNUFFT(np.complex64, (n, n, n), m=M8, sigma=2.0, kernel_evalmode=nufft.FastApproximation(), backend=nufft.ROCBackend(0))
    info = plan.info()
    assert info.spread_method == 2 and info.patch_f32acc == 1
    oplan = O.OraclePlan((n, n, n), is_real=False, dtype=np.float64, coord_dtype=np.float32, M=M8, sigma=2.0, evalmode=O.FAST_APPROXIMATION)
    nufft.set_points(plan, tuple(torch.from_numpy(x).cuda() for x in xs))
    assert plan.spread_engine_used() == "mfma_patches" and plan.interp_engine_used() == "marching_ring"
    O.set_points(oplan, xs)
    u = torch.empty(plan.shape, dtype=torch.complex64, device="cuda")
    nufft.exec_type1(u, plan, torch.from_numpy(v).cuda())
    ref = CO.exec_type1(oplan, v.astype(np.complex128))
    e1 = float(np.linalg.norm(u.cpu().numpy().astype(np.complex128) - ref) / np.linalg.norm(ref))
    assert e1 < 1e-5, e1
    w = (rng.standard_normal(plan.shape) + 1j * rng.standard_normal(plan.shape)).astype(np.complex64)
    out = torch.empty(Np, dtype=torch.complex64, device="cuda")
    nufft.exec_type2(out, plan, torch.from_numpy(w).cuda())
    ref2 = CO.exec_type2(oplan, w.astype(np.complex128))
    e2 = float(np.linalg.norm(out.cpu().numpy().astype(np.complex128) - ref2) / np.linalg.norm(ref2))
    assert e2 < 1e-5, e2


@pytest.mark.parametrize("Z,M,engine", [
    (np.complex128, 4, "mfma_patches"), (np.float64, 6, "mfma_patches"), (np.complex64, 8, "mfma_patches"),
    (np.float64, 4, "mfma_patches"), (np.complex64, 8, "lds_tiles"), (np.float32, 4, "mfma_patches"),
    (np.float64, 4, "marching_ring"), (np.float32, 4, "marching_ring"), (np.complex128, 3, "marching_ring"), (np.float64, 6, "marching_ring"),
])
def test_dense_point_sets_match_c_oracle(Z, M, engine):
    """Dense sets (oversampled 128^3 = 32^3 bins, Np = 2e6: 61 points per bin, every K-batch of the patch engine full and
    every chunk of a run at capacity) against the C oracle — full rel-L2 over all output modes (type 1) and all points
    (type 2), at the reference's bounds (test/pseudo_gpu.jl:159-171: 1e-7 Float64, 1e-5 Float32).  Float32 plans are
    compared with the Float64 C oracle that locates the points in Float32 (coord_dtype), since the reference's
    un-normalised Float32 window overflows at (3-D, M = 8)."""
    from nufft_pkg import nufft
    from oracle import nufft_oracle as O, c_oracle as CO
    if not CO.available():
        pytest.skip("C oracle not built")
    Zt = np.dtype(Z)
    is_real = Zt.kind == "f"
    T = np.float32 if Zt in (np.dtype(np.float32), np.dtype(np.complex64)) else np.float64
    n, Np = 64, 2_000_000
    rng = np.random.default_rng(1000 + M)
    xs = [(rng.random(Np) * O.TWO_PI).astype(T) for _ in range(3)]
    v = (rng.standard_normal(Np) if is_real else rng.standard_normal(Np) + 1j * rng.standard_normal(Np)).astype(Zt)
    plan = nufft.PlanNUFFT(Zt, (n, n, n), m=M, sigma=2.0, kernel_evalmode=nufft.FastApproximation(), spread_method=engine,
                           backend=nufft.ROCBackend(0))
    oplan = O.OraclePlan((n, n, n), is_real=is_real, dtype=np.float64, coord_dtype=(np.float32 if T == np.float32 else None),
                         M=M, sigma=2.0, evalmode=O.FAST_APPROXIMATION)
    nufft.set_points(plan, tuple(torch.from_numpy(x).cuda() for x in xs))
    # (61 points per bin: on plans of the marching window set_points hands such a set to the window's dense-set engine — the matrix pipe
    # accumulates a bin in registers, DESIGN.md section 4.12 — from 24 (m = 4) / 6 (m = 6) points per bin on with the polynomial window)
    assert plan.spread_engine_used() == (engine + "_dense" if engine == "marching_ring" and M >= 4 else engine)
    O.set_points(oplan, xs)
    wide = np.float64 if is_real else np.complex128
    u = torch.empty(plan.shape, dtype=plan.eltype, device="cuda")
    nufft.exec_type1(u, plan, torch.from_numpy(v).cuda())
    ref = CO.exec_type1(oplan, v.astype(wide))
    tol = 1e-5 if T == np.float32 else 1e-7
    e1 = float(np.linalg.norm(u.cpu().numpy().astype(np.complex128) - ref) / np.linalg.norm(ref))
    assert e1 < tol, e1
    w = (rng.standard_normal(plan.shape) + 1j * rng.standard_normal(plan.shape)).astype(np.complex64 if T == np.float32 else np.complex128)
    out = torch.empty(Np, dtype=plan.Z, device="cuda")
    nufft.exec_type2(out, plan, torch.from_numpy(w).cuda())
    ref2 = CO.exec_type2(oplan, w.astype(np.complex128))
    e2 = float(np.linalg.norm(out.cpu().numpy().astype(wide) - ref2) / np.linalg.norm(ref2))
    assert e2 < tol, e2
