"""Consistency tests of the HIP path against ITSELF (different engines / code paths of the library must produce the same
sums).  These are NOT parity tests — nothing here involves the oracle; the oracle-backed counterparts are in
tests/test_gpu_parity.py (test_callbacks_match_oracle, test_both_spreading_engines_match_oracle,
test_*_every_instantiation, test_dense_point_sets_match_c_oracle in tests/test_gpu_fullsize.py).  Kept because they compare
at round-off level (1e-12 ... 1e-16), far below the oracle tolerances, and catch an engine that drifts."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import nufft_oracle as O  # noqa: E402,F401
from test_gpu_parity import _nufft, _make_case, _rel, _rtol, _oracle_inputs, plan_real_dtype  # noqa: E402,F401  (shared helpers)

@pytest.mark.parametrize("Z,Ns,C", [(np.float32, (64, 32, 16), 1), (np.complex64, (64, 32, 16), 1), (np.complex128, (64, 32, 32), 2),
                                    (np.float64, (32, 32, 16), 2), (np.complex128, (40, 24), 1), (np.float64, (128,), 1)])
def test_callbacks_menu(Z, Ns, C):
    """test/callbacks.jl:6-66: random per-point weights as the non-uniform callback and 1/k² (0 at k = 0) as
    the uniform callback; the reference result applies the same functions before / after plain transforms.
    Same sizes as the reference for the two Float32 cases (Ns = (64, 32, 16), Np = prod(Ns) ÷ 3); the others
    cover the pruned-FFT path (power-of-two oversampled grid), ntransforms = 2, 2-D and 1-D."""
    nufft = _nufft()
    Zt = np.dtype(Z)
    is_real = Zt.kind == "f"
    T = np.float32 if Zt in (np.dtype(np.float32), np.dtype(np.complex64)) else np.float64
    Np = int(np.prod(Ns)) // 3
    rng = np.random.default_rng(42)
    weights = rng.random(Np).astype(T)
    ks = [(np.fft.rfftfreq(N, 1 / N) if (d == 0 and is_real) else np.fft.fftfreq(N, 1 / N)) for d, N in enumerate(Ns)]
    k2 = sum(np.reshape(k ** 2, [-1 if e == d else 1 for e in range(len(Ns))][::-1]) for d, k in enumerate(ks))
    factors = np.where(k2 == 0, 0.0, 1.0 / np.where(k2 == 0, 1.0, k2)).astype(T)     # reversed axes = torch layout
    xs = [(rng.random(Np) * 2 * np.pi).astype(T) for _ in Ns]
    vs = [(rng.standard_normal(Np) if is_real else rng.standard_normal(Np) + 1j * rng.standard_normal(Np)).astype(Zt)
          for _ in range(C)]
    plan = nufft.PlanNUFFT(Zt, Ns, ntransforms=C, backend=nufft.ROCBackend(0))        # default parameters, as the reference
    dev = plan.device
    xd = tuple(torch.from_numpy(x).to(dev) for x in xs)
    nufft.set_points(plan, xd)
    wd, fd = torch.from_numpy(weights).to(dev), torch.from_numpy(np.ascontiguousarray(factors)).to(dev)
    assert tuple(fd.shape) == plan.shape
    cb = nufft.NUFFTCallbacks(nonuniform=nufft.PointWeights(wd), uniform=nufft.ModeFactors(fd))
    tup = (lambda t: t if C > 1 else t[0])

    # reference: callbacks applied outside plain transforms (test/callbacks.jl:36-47)
    vd = tuple(torch.from_numpy(v).to(dev) for v in vs)
    t1_in = tuple(v * wd for v in vd)
    t1_ref = tuple(torch.empty(plan.shape, dtype=plan.eltype, device=dev) for _ in range(C))
    nufft.exec_type1(tup(t1_ref), plan, tup(t1_in))
    t1_ref = tuple(u * fd for u in t1_ref)
    t2_in = tuple(u * fd for u in t1_ref)
    t2_ref = tuple(torch.empty(Np, dtype=plan.Z, device=dev) for _ in range(C))
    nufft.exec_type2(tup(t2_ref), plan, tup(t2_in))
    t2_ref = tuple(v * wd for v in t2_ref)

    # fused callbacks
    ws = tuple(torch.empty(plan.shape, dtype=plan.eltype, device=dev) for _ in range(C))
    nufft.exec_type1(tup(ws), plan, tup(vd), callbacks=cb)
    wp = tuple(torch.empty(Np, dtype=plan.Z, device=dev) for _ in range(C))
    nufft.exec_type2(tup(wp), plan, tup(ws), callbacks=cb)
    tol = 1e-5 if T == np.float32 else 1e-12              # `≈` in the reference: rtol = sqrt(eps)
    for c in range(C):
        assert _rel(ws[c].cpu().numpy(), t1_ref[c].cpu().numpy()) < tol
        assert _rel(wp[c].cpu().numpy(), t2_ref[c].cpu().numpy()) < tol
    # only one of the two, and argument checks
    only_w = nufft.NUFFTCallbacks(nonuniform=nufft.PointWeights(wd))
    w2 = tuple(torch.empty(plan.shape, dtype=plan.eltype, device=dev) for _ in range(C))
    nufft.exec_type1(tup(w2), plan, tup(vd), callbacks=only_w)
    plain = tuple(torch.empty(plan.shape, dtype=plan.eltype, device=dev) for _ in range(C))
    nufft.exec_type1(tup(plain), plan, tup(t1_in))
    assert _rel(w2[0].cpu().numpy(), plain[0].cpu().numpy()) < tol
    with pytest.raises(NotImplementedError):
        nufft.NUFFTCallbacks(nonuniform=lambda v, n: v)
    with pytest.raises(nufft.DimensionMismatch):
        nufft.exec_type1(tup(w2), plan, tup(vd), callbacks=nufft.NUFFTCallbacks(nonuniform=nufft.PointWeights(wd[:-1].contiguous())))


@pytest.mark.parametrize("Z,n,M,Np,C,dist", [
    (torch.float64, 64, 4, 1_000_000, 1, "uniform"),       # 30 points per bin: several K-batches per chunk, several chunks per run
    (torch.float64, 64, 4, 200_000, 1, "cluster"),
    (torch.complex64, 64, 8, 1_000_000, 1, "uniform"),
    (torch.complex128, 64, 6, 500_000, 2, "uniform"),
    (torch.float32, 96, 5, 1_000_000, 1, "cluster"),
    (torch.float64, 64, 10, 300_000, 1, "uniform"),
])
def test_spreading_engines_agree_on_dense_point_sets(Z, n, M, Np, C, dist):
    """The oracle-sized cases above hold at most a point or two per bin.  Here the two engines (independent
    implementations: LDS atomics vs matrix-pipe accumulation in registers) must agree on dense and clustered sets,
    where a run of the sorted array spans many chunks and every K-batch of the patches is full."""
    nufft = _nufft()
    T = torch.float32 if Z in (torch.float32, torch.complex64) else torch.float64
    g = torch.Generator(device="cuda").manual_seed(3)
    if dist == "uniform":
        xs = tuple(torch.rand(Np, dtype=T, device="cuda", generator=g) * (2 * np.pi) for _ in range(3))
    else:
        xs = tuple(torch.randn(Np, dtype=T, device="cuda", generator=g) * 0.3 + np.pi for _ in range(3))
    vs = tuple(torch.randn(Np, dtype=Z, device="cuda", generator=g) for _ in range(C))
    outs = []
    for eng in ("lds_tiles", "mfma_patches"):
        p = nufft.PlanNUFFT(Z, (n, n, n), m=M, ntransforms=C, spread_method=eng, kernel_evalmode=nufft.FastApproximation(),
                            backend=nufft.ROCBackend(0))
        nufft.set_points(p, xs)
        us = tuple(torch.empty(p.shape, dtype=p.eltype, device="cuda") for _ in range(C))
        nufft.exec_type1(us if C > 1 else us[0], p, vs if C > 1 else vs[0])
        outs.append(us)
    tol = 5e-6 if T == torch.float32 else 1e-13
    for c in range(C):
        assert bool(torch.isfinite(torch.view_as_real(outs[1][c])).all())
        assert float((outs[0][c] - outs[1][c]).norm() / outs[0][c].norm()) < tol


def test_cube_accumulation_variant_of_the_tile_kernel(monkeypatch):
    """NUFFT_SPREAD_CUBES=1 (opt-in): the LDS-tile kernel accumulates four points at a time cube by
    cube (v_mfma_f64_4x4x4 + one ds_add_f64 per cube) instead of plane by plane.  Same sums: must agree with the default
    face mapping on sparse, dense and one-cell point sets (partially filled K-batches, stencils two cubes below a tile)."""
    nufft = _nufft()
    plans = {}
    for c in ("0", "1"):
        monkeypatch.setenv("NUFFT_SPREAD_CUBES", c)
        plans[c] = nufft.PlanNUFFT(torch.float64, (64, 64, 64), m=4, sigma=2.0, spread_method="lds_tiles",
                                   kernel_evalmode=nufft.FastApproximation(), backend=nufft.ROCBackend(0))
    g = torch.Generator(device="cuda").manual_seed(5)
    cases = [tuple(torch.rand(n, dtype=torch.float64, device="cuda", generator=g) * (2 * np.pi) for _ in range(3)) for n in (3000, 400_000)]
    for n in (5, 6, 13):                                  # n points in one cell
        cases.append(tuple((torch.full((n,), 70.3 / 128 * 2 * np.pi, dtype=torch.float64, device="cuda")
                            + 1e-3 * torch.rand(n, dtype=torch.float64, device="cuda", generator=g)).contiguous() for _ in range(3)))
    for xs in cases:
        v = torch.randn(xs[0].numel(), dtype=torch.float64, device="cuda", generator=g)
        outs = []
        for c in ("0", "1"):
            nufft.set_points(plans[c], xs)
            u = torch.empty(plans[c].shape, dtype=torch.complex128, device="cuda")
            nufft.exec_type1(u, plans[c], v)
            outs.append(u)
        assert float((outs[0] - outs[1]).norm() / outs[0].norm()) < 1e-13


@pytest.mark.parametrize("Z", [np.float64, np.complex64])
def test_deferred_spread_stage_completes_the_grid_wherever_it_is_consumed(Z, monkeypatch):
    """nufft_spread_deferred (the spreading stage as exec_type1 enqueues it: on the marching window's halo variant the stencil reach
    beyond the columns sits in a side buffer until a consumer of the grid adds it).  Every consumer must see the complete grid:
    nufft_copy_grid, nufft_interpolate, and the FFT stage (stage by stage = exec_type1); an abandoned deferred spread must not leak
    into a later transform (exec_type2 overwrites the grids; set_points voids it)."""
    import ctypes as C
    monkeypatch.setenv("NUFFT_SMARCH_HALO", "2")
    dims, M, Np = (64, 64, 32), 3, 30000
    nufft, plan, oplan, xs, vs = _make_case(Z, dims, M, 2.0, O.FAST_APPROXIMATION, 1, Np, seed=9, spread_method="marching_ring")
    from nonuniformffts_jl_amd.plan import _check, _ptr_table      # (the package's import name, registered by nufft_pkg)
    assert plan.info().ring_halo == 1
    lib, dev = nufft.lib, plan.device
    tol = 1e-13 if np.dtype(Z) == np.float64 else 2e-6          # (LDS float atomics: the order of the sums differs from launch to launch)

    def same(a, b):
        return float((a - b).norm() / b.norm()) < tol
    xd = tuple(torch.from_numpy(x).to(dev) for x in xs)
    vd = torch.from_numpy(vs[0]).to(dev)
    nufft.set_points(plan, xd)
    s = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    # the complete grid, from the self-contained stage
    nufft.spread_from_points(plan, vd)
    full = nufft.oversampled_grid(plan).clone()
    # deferred, then read back: the copy completes it
    _check(lib.nufft_spread_deferred(plan._handle, _ptr_table((vd,)), s))
    assert same(nufft.oversampled_grid(plan), full)
    # deferred, then interpolate from the grid: the same values as from the complete grid
    out_ref = torch.empty(Np, dtype=plan.Z, device=dev)
    nufft.spread_from_points(plan, vd)
    nufft.interpolate(plan, out_ref)
    out = torch.empty_like(out_ref)
    _check(lib.nufft_spread_deferred(plan._handle, _ptr_table((vd,)), s))
    nufft.interpolate(plan, out)
    assert same(out, out_ref)
    # deferred + FFT + deconvolution, stage by stage = exec_type1
    u_ref = torch.empty(plan.shape, dtype=plan.eltype, device=dev)
    nufft.exec_type1(u_ref, plan, vd)
    u = torch.empty_like(u_ref)
    _check(lib.nufft_spread_deferred(plan._handle, _ptr_table((vd,)), s))
    _check(lib.nufft_fft_forward(plan._handle, s))
    _check(lib.nufft_deconvolve_truncate(plan._handle, _ptr_table((u,)), s))
    assert same(u, u_ref)
    # an abandoned deferred spread does not leak into a type-2 transform, nor survive set_points
    w = torch.empty(Np, dtype=plan.Z, device=dev)
    nufft.exec_type2(w, plan, u_ref)
    _check(lib.nufft_spread_deferred(plan._handle, _ptr_table((vd,)), s))
    w2 = torch.empty_like(w)
    nufft.exec_type2(w2, plan, u_ref)
    assert same(w2, w)
    _check(lib.nufft_spread_deferred(plan._handle, _ptr_table((vd,)), s))
    nufft.set_points(plan, xd)
    nufft.spread_from_points(plan, vd)
    assert same(nufft.oversampled_grid(plan), full)
