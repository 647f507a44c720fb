"""hipGraph capture of the hot path.

set_points! / exec_type1! / exec_type2! only enqueue work on the stream they are given (no allocation once the plan has
seen a point set of that size, no host read-back, no synchronisation), so the whole sequence can be captured into a
hipGraph once and replayed on new data in the same buffers — what a launch-bound caller (many small transforms) wants.
The reference has no counterpart (KernelAbstractions launches eagerly); the parity bar is the same as for eager calls:
GPU vs CPU oracle, rtol 1e-7 (Float64) / 1e-5 (Float32) on the 2-norm (test/pseudo_gpu.jl:159-171).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import nufft_oracle as O  # noqa: E402


def _rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.linalg.norm((a - b).ravel()) / np.linalg.norm(b.ravel()))


@pytest.mark.parametrize("Z,dims,sigma,evalmode", [
    (np.float64, (256,), 2.0, O.DIRECT),                     # C1 shape: rocFFT inside the captured sequence
    (np.float64, (48, 40), 1.5, O.FAST_APPROXIMATION),       # general path (rocFFT + deconvolution kernels)
    (np.float64, (32, 32, 32), 2.0, O.FAST_APPROXIMATION),   # pruned FFT passes
    (np.complex64, (32, 64, 32), 2.0, O.DIRECT),
    (np.float64, (64, 64, 32), 2.0, O.FAST_APPROXIMATION),   # spreading window in its halo variant: the captured FFT pass adds the side buffer
])
def test_graph_replay_matches_oracle_on_new_data(Z, dims, sigma, evalmode):
    from nufft_pkg import nufft
    Zt = np.dtype(Z)
    is_real = Zt.kind == "f"
    T = np.float32 if Zt in (np.dtype(np.float32), np.dtype(np.complex64)) else np.float64
    Np = 3000
    mode = nufft.Direct() if evalmode == O.DIRECT else nufft.FastApproximation()
    plan = nufft.PlanNUFFT(Zt, dims, m=4, sigma=sigma, kernel_evalmode=mode, backend=nufft.ROCBackend(0))
    oplan = O.OraclePlan(dims, is_real=is_real, dtype=T, M=4, sigma=sigma, evalmode=evalmode)
    dev = plan.device
    if dims == (64, 64, 32):
        assert plan.info().spread_method == 3 and plan.info().ring_halo == 1

    def inputs(seed):
        rng = np.random.default_rng(seed)
        xs = [((rng.random(Np) * 3 - 1) * O.TWO_PI).astype(T) for _ in dims]
        v = rng.standard_normal(Np) if is_real else rng.standard_normal(Np) + 1j * rng.standard_normal(Np)
        return xs, v.astype(Zt)

    xs0, v0 = inputs(1)
    xd = tuple(torch.from_numpy(x).to(dev) for x in xs0)
    vd = torch.from_numpy(v0).to(dev)
    ud = torch.empty(plan.shape, dtype=plan.eltype, device=dev)
    wd = torch.empty(Np, dtype=plan.Z, device=dev)

    def step():
        nufft.set_points(plan, xd)
        nufft.exec_type1(ud, plan, vd)
        nufft.exec_type2(wd, plan, ud)       # type 2 of the type-1 result: the graph chains both transforms

    step()                                   # sizes the plan's point buffers (allocation is not capturable)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()

    tol = 1e-7 if T == np.float64 else 1e-5
    for seed in (2, 3):
        xs, v = inputs(seed)
        for d in range(len(dims)):
            xd[d].copy_(torch.from_numpy(xs[d]))     # eager work between replays (new data, same buffers)
        vd.copy_(torch.from_numpy(v))
        ud.zero_(); wd.zero_()
        graph.replay()
        torch.cuda.synchronize()
        O.set_points(oplan, xs)
        ref1 = O.exec_type1(oplan, v)
        assert _rel(ud.cpu().numpy(), ref1) < tol
        ref2 = O.exec_type2(oplan, ref1.astype(np.complex64 if T == np.float32 else np.complex128))
        assert _rel(wd.cpu().numpy(), ref2) < tol
    # many replays back to back, then an eager call on the same plan
    for _ in range(20):
        graph.replay()
    torch.cuda.synchronize()
    assert _rel(wd.cpu().numpy(), ref2) < tol
    step()
    torch.cuda.synchronize()
    assert _rel(wd.cpu().numpy(), ref2) < tol


def test_graph_replay_on_a_plan_of_the_column_layer_sort():
    """A captured set_points! + exec_type1! + exec_type2! on a plan whose points are sorted by column layers (binsort.hip, CoarseSort): both
    sorts and all three interpolation kernels are part of the captured sequence, the device flags of every replay pick what runs — replayed on
    a uniform set (column-layer sort, staged ring), then on a set concentrated in a corner (fine sort, tile kernels), then uniform again."""
    from nufft_pkg import nufft
    dims, Np = (256, 256, 32), 120000
    plan = nufft.PlanNUFFT(np.float64, dims, m=4, sigma=2.0, kernel_evalmode=nufft.Direct(), backend=nufft.ROCBackend(0))
    info = plan.info()
    assert info.spread_method == 3 and info.ring_halo == 1 and info.sort_column[0] > 0
    oplan = O.OraclePlan(dims, is_real=True, dtype=np.float64, M=4, sigma=2.0, evalmode=O.DIRECT)
    dev = plan.device
    rng = np.random.default_rng(21)

    def inputs(kind):
        xs = [rng.random(Np) * O.TWO_PI for _ in dims]
        if kind == "corner":
            xs = [0.05 * x for x in xs]
        return xs, rng.standard_normal(Np)

    xs0, v0 = inputs("uniform")
    xd = tuple(torch.from_numpy(x).to(dev) for x in xs0)
    vd = torch.from_numpy(v0).to(dev)
    ud = torch.empty(plan.shape, dtype=plan.eltype, device=dev)
    wd = torch.empty(Np, dtype=plan.Z, device=dev)

    def step():
        nufft.set_points(plan, xd)
        nufft.exec_type1(ud, plan, vd)
        nufft.exec_type2(wd, plan, ud)

    step()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    for kind in ("uniform", "corner", "uniform"):
        xs, v = inputs(kind)
        for d in range(3):
            xd[d].copy_(torch.from_numpy(xs[d]))
        vd.copy_(torch.from_numpy(v))
        ud.zero_(); wd.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert plan.sort_columns_used() == (kind == "uniform"), kind
        O.set_points(oplan, xs)
        ref1 = O.exec_type1(oplan, v)
        assert _rel(ud.cpu().numpy(), ref1) < 1e-7, kind
        assert _rel(wd.cpu().numpy(), O.exec_type2(oplan, ref1)) < 1e-7, kind


def test_graph_replay_on_a_plan_of_the_slab_sort():
    """A captured set_points! + exec_type1! + exec_type2! on a plan that orders its points by fine bins in two levels (binsort.hip, slab sort):
    both sorts are part of the captured sequence and the fullest slab of every replay decides which one runs — replayed on a uniform set
    (slabs), on a tight cluster (global atomics), and on a uniform set again."""
    from nufft_pkg import nufft
    dims, Np = (48, 40, 36), 40000
    plan = nufft.PlanNUFFT(np.complex64, dims, m=5, sigma=2.0, kernel_evalmode=nufft.Direct(), backend=nufft.ROCBackend(0))
    assert plan.info().sort_column[0] == 0
    oplan = O.OraclePlan(dims, is_real=False, dtype=np.float32, M=5, sigma=2.0, evalmode=O.DIRECT)
    dev = plan.device
    rng = np.random.default_rng(22)

    def inputs(kind):
        if kind == "cluster":
            xs = [(1.0 + 0.01 * rng.standard_normal(Np)).astype(np.float32) for _ in dims]
        else:
            xs = [(rng.random(Np) * O.TWO_PI).astype(np.float32) for _ in dims]
        return xs, (rng.standard_normal(Np) + 1j * rng.standard_normal(Np)).astype(np.complex64)

    xs0, v0 = inputs("uniform")
    xd = tuple(torch.from_numpy(x).to(dev) for x in xs0)
    vd = torch.from_numpy(v0).to(dev)
    ud = torch.empty(plan.shape, dtype=plan.eltype, device=dev)
    wd = torch.empty(Np, dtype=plan.Z, device=dev)

    def step():
        nufft.set_points(plan, xd)
        nufft.exec_type1(ud, plan, vd)
        nufft.exec_type2(wd, plan, ud)

    step()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    for kind in ("uniform", "cluster", "uniform"):
        xs, v = inputs(kind)
        for d in range(3):
            xd[d].copy_(torch.from_numpy(xs[d]))
        vd.copy_(torch.from_numpy(v))
        ud.zero_(); wd.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert plan.sort_method_used() == ("slabs" if kind == "uniform" else "fine_bins"), kind
        O.set_points(oplan, xs)
        ref1 = O.exec_type1(oplan, v)
        assert _rel(ud.cpu().numpy(), ref1) < 2e-5, kind
        assert _rel(wd.cpu().numpy(), O.exec_type2(oplan, ref1)) < 2e-5, kind


@pytest.mark.parametrize("consumer", ["eager_fft", "eager_interpolate", "second_graph", "eager_copy", "voided_by_type2"])
def test_deferred_spread_captured_alone_is_consumed_correctly(consumer):
    """The state "side buffer written, stencil reach not yet added to `us`" of the spreading window's halo variant is a device word that
    the spreading kernel sets and its consumers clear (plan.cpp: halo_hint / kHaloStateWord), not host state: a nufft_spread_deferred
    captured ALONE in a hipGraph and replayed later is completed by whatever consumes the grid next — an eager FFT, an eager
    interpolation, a copy, another graph — exactly once; and a type-2 transform in between voids it."""
    import ctypes as Ct
    from nufft_pkg import nufft
    from nonuniformffts_jl_amd.plan import _check, _ptr_table
    dims, Np, M = (64, 48, 64), 6000, 4
    plan = nufft.PlanNUFFT(np.float64, dims, m=M, sigma=2.0, kernel_evalmode=nufft.Direct(), spread_method="marching_ring", backend=nufft.ROCBackend(0))
    info = plan.info()
    assert info.spread_method == 3 and info.ring_halo == 1
    oplan = O.OraclePlan(dims, is_real=True, dtype=np.float64, M=M, sigma=2.0, evalmode=O.DIRECT)
    dev = plan.device
    lib, h = nufft.lib, plan._handle
    rng = np.random.default_rng(77)
    xs = [rng.random(Np) * O.TWO_PI for _ in dims]
    v1, v2 = rng.standard_normal(Np), rng.standard_normal(Np)
    xd = tuple(torch.from_numpy(x).to(dev) for x in xs)
    vd = torch.from_numpy(v1).to(dev)
    ud = torch.empty(plan.shape, dtype=plan.eltype, device=dev)
    wd = torch.empty(Np, dtype=torch.float64, device=dev)
    nufft.set_points(plan, xd)
    O.set_points(oplan, xs)
    nufft.exec_type1(ud, plan, vd)          # warm-up: everything allocated, an eager deferred spread consumed by the fused pass
    torch.cuda.synchronize()
    scale = 2.0 ** sum(info.window_scale_log2[d] for d in range(3))

    def stream():
        return Ct.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    g_spread = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g_spread):
        _check(lib.nufft_spread_deferred(h, _ptr_table((vd,)), stream()))
    vd.copy_(torch.from_numpy(v2))          # new values: the replay spreads v2
    ref_grid = O.spread(oplan, [v2])[0]
    ref1 = O.exec_type1(oplan, v2)

    if consumer == "eager_fft":
        g_spread.replay()
        _check(lib.nufft_fft_forward(h, stream()))
        _check(lib.nufft_deconvolve_truncate(h, _ptr_table((ud,)), stream()))
        torch.cuda.synchronize()
        assert _rel(ud.cpu().numpy(), ref1) < 1e-7
        # `us` still lacks the reach after the fused pass: a copy completes it, once
        g = nufft.oversampled_grid(plan, 0).cpu().numpy() / scale
        assert _rel(g, ref_grid) < 1e-12
        assert np.array_equal(nufft.oversampled_grid(plan, 0).cpu().numpy() / scale, g)
    elif consumer == "eager_interpolate":
        g_spread.replay()
        nufft.interpolate(plan, wd)
        torch.cuda.synchronize()
        # (device windows carry the factor 2^k in both directions: DESIGN.md section 2)
        assert _rel(wd.cpu().numpy() / scale ** 2, O.interpolate(oplan, [ref_grid.copy()])[0]) < 1e-7
    elif consumer == "eager_copy":
        for _ in range(3):                  # replays back to back: each spread overwrites grid and side buffer, nothing accumulates
            g_spread.replay()
        g = nufft.oversampled_grid(plan, 0).cpu().numpy() / scale
        assert _rel(g, ref_grid) < 1e-12
    elif consumer == "second_graph":
        g_rest = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_rest):
            _check(lib.nufft_fft_forward(h, stream()))
            _check(lib.nufft_deconvolve_truncate(h, _ptr_table((ud,)), stream()))
        for _ in range(2):
            ud.zero_()
            g_spread.replay()
            g_rest.replay()
            torch.cuda.synchronize()
            assert _rel(ud.cpu().numpy(), ref1) < 1e-7
        # the second graph on a grid that an eager, complete nufft_spread left (nothing pending): the reach must not be added twice
        _check(lib.nufft_spread(h, _ptr_table((vd,)), stream()))
        ud.zero_()
        g_rest.replay()
        torch.cuda.synchronize()
        assert _rel(ud.cpu().numpy(), ref1) < 1e-7
    else:
        # a type-2 transform overwrites the grids: the pending side buffer of the replayed spread is void, not added to the new field
        g_spread.replay()
        spec = torch.from_numpy(ref1).to(dev)
        nufft.exec_type2(wd, plan, spec)
        torch.cuda.synchronize()
        assert _rel(wd.cpu().numpy(), O.exec_type2(oplan, ref1)) < 1e-7
        g = nufft.oversampled_grid(plan, 0)          # (a copy after it must find nothing to add either)
        nufft.interpolate(plan, wd)
        torch.cuda.synchronize()
        assert _rel(wd.cpu().numpy(), O.exec_type2(oplan, ref1)) < 1e-7
        del g
