"""Plan lifetime, stream and reuse behaviour of the HIP path (no reference counterpart: the reference relies on
Julia's GC and KernelAbstractions streams; these are the conventions INTEGRATION.md promises the Julia shim)."""
import gc

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _case(nufft, Z=torch.float64, dims=(48, 40, 36), Np=30000, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    xs = tuple(torch.rand(Np, dtype=torch.float64, device="cuda", generator=g) * (2 * np.pi) for _ in dims)
    v = torch.randn(Np, dtype=torch.float64, device="cuda", generator=g)
    return xs, v


def test_plan_destroy_returns_device_memory():
    from nufft_pkg import nufft
    torch.cuda.synchronize()
    xs, v = _case(nufft)
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(12):
        p = nufft.PlanNUFFT(torch.float64, (128, 128, 128), backend=nufft.ROCBackend(0))
        nufft.set_points(p, tuple(x[:1000].contiguous() for x in xs))
        assert p.info().workspace_bytes > 128 ** 3 * 8
        del p
        gc.collect()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 64 << 20          # 12 plans of ~0.3 GB each would show if the handle leaked


def test_side_stream_and_interleaved_plans():
    """All work is enqueued on the stream current at call time; two plans with different parameters can be
    used alternately; a plan can take a larger and then a smaller point set."""
    from nufft_pkg import nufft
    dims = (48, 40, 36)
    xs, v = _case(nufft, dims=dims)
    pa = nufft.PlanNUFFT(torch.float64, dims, m=4, backend=nufft.ROCBackend(0))
    pb = nufft.PlanNUFFT(torch.float64, dims, m=6, sigma=1.5, kernel=nufft.KaiserBesselKernel(), backend=nufft.ROCBackend(0))
    ua = torch.empty(pa.shape, dtype=torch.complex128, device="cuda")
    ub = torch.empty(pb.shape, dtype=torch.complex128, device="cuda")
    nufft.set_points(pa, xs); nufft.set_points(pb, xs)
    nufft.exec_type1(ua, pa, v); nufft.exec_type1(ub, pb, v)
    torch.cuda.synchronize()
    ref_a, ref_b = ua.clone(), ub.clone()
    assert float((ua - ub).norm() / ua.norm()) < 1e-6          # same transform, two accuracies
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ua.zero_(); ub.zero_()
        nufft.set_points(pb, xs)
        nufft.exec_type1(ub, pb, v)
        nufft.set_points(pa, xs)
        nufft.exec_type1(ua, pa, v)
    side.synchronize()
    assert float((ua - ref_a).norm() / ref_a.norm()) < 1e-12
    assert float((ub - ref_b).norm() / ref_b.norm()) < 1e-12
    # grow, then shrink the point set of one plan
    g = torch.Generator(device="cuda").manual_seed(5)
    big = tuple(torch.rand(200000, dtype=torch.float64, device="cuda", generator=g) * (2 * np.pi) for _ in dims)
    vb = torch.randn(200000, dtype=torch.float64, device="cuda", generator=g)
    nufft.set_points(pa, big)
    nufft.exec_type1(ua, pa, vb)
    k = torch.tensor([3.0, -2.0, 5.0], dtype=torch.float64, device="cuda")
    exact = (vb * torch.polar(torch.ones_like(vb), -(k[0] * big[0] + k[1] * big[1] + k[2] * big[2]))).sum()
    assert float((ua[5, 38, 3] - exact).abs() / exact.abs()) < 1e-6
    nufft.set_points(pa, xs)
    nufft.exec_type1(ua, pa, v)
    assert float((ua - ref_a).norm() / ref_a.norm()) < 1e-12


def test_a_later_plan_does_not_lower_an_earlier_plans_lds_allowance(monkeypatch):
    """hipFuncAttributeMaxDynamicSharedMemorySize belongs to the kernel, not to the plan: a plan with a small tile (or few sort keys)
    created after a plan with a large one — same kernel instantiation — must leave the earlier plan launchable.  2-D tile kernels with
    a large and a tiny tile; and the LDS histograms of the two sorts: a slab-sorted plan with 34 560 keys (135 KiB), then a column-layer
    plan (fewer keys), then the first plan's set_points."""
    from nufft_pkg import nufft
    g = torch.Generator(device="cuda").manual_seed(9)
    big = nufft.PlanNUFFT(torch.float64, (256, 256), m=4, backend=nufft.ROCBackend(0))
    small = nufft.PlanNUFFT(torch.float64, (16, 16), m=4, backend=nufft.ROCBackend(0))
    assert big.info().lds_bytes_spread > small.info().lds_bytes_spread
    xs = tuple(torch.rand(20000, dtype=torch.float64, device="cuda", generator=g) * (2 * np.pi) for _ in range(2))
    v = torch.randn(20000, dtype=torch.float64, device="cuda", generator=g)
    k = torch.tensor([7.0, -11.0], dtype=torch.float64, device="cuda")
    exact = (v * torch.polar(torch.ones_like(v), -(k[0] * xs[0] + k[1] * xs[1]))).sum()
    for p in (small, big):
        u = torch.empty(p.shape, dtype=torch.complex128, device="cuda")
        nufft.set_points(p, xs)
        nufft.exec_type1(u, p, v)
        if p is big:
            assert float((u[256 - 11, 7] - exact).abs() / exact.abs()) < 1e-6     # (axes reversed: [k2, k1])
    # sorts: slabs of one bin row on a 64 x 768 x 720 grid = 180 x 192 keys (a small NUFFT_SLAB_FILL keeps the slabs that low)
    monkeypatch.setenv("NUFFT_SLAB_FILL", "5")
    Np = 8_000_000
    xs3 = tuple(torch.rand(Np, dtype=torch.float32, device="cuda", generator=g) * (2 * np.pi) for _ in range(3))
    v3 = torch.randn(Np, dtype=torch.float32, device="cuda", generator=g)
    slab = nufft.PlanNUFFT(torch.float32, (32, 384, 360), m=4, backend=nufft.ROCBackend(0))
    col = nufft.PlanNUFFT(torch.float64, (256, 256, 32), m=4, backend=nufft.ROCBackend(0))
    nufft.set_points(col, tuple(x[:100000].double() for x in xs3))
    nufft.set_points(slab, xs3)
    assert slab.sort_method_used() == "slabs"
    u3 = torch.empty(slab.shape, dtype=torch.complex64, device="cuda")
    nufft.exec_type1(u3, slab, v3)
    k3 = torch.tensor([3.0, -2.0, 5.0], dtype=torch.float64, device="cuda")
    x64 = [x.double() for x in xs3]
    exact3 = (v3.double() * torch.polar(torch.ones_like(x64[0]), -(k3[0] * x64[0] + k3[1] * x64[1] + k3[2] * x64[2]))).sum()
    assert float((u3[5, 384 - 2, 3].to(torch.complex128) - exact3).abs() / exact3.abs()) < 2e-3


@pytest.mark.parametrize("name,Z,n,Np,M,C,limit", [("C2", np.float64, 256, 10_000_000, 4, 1, 3.8e9), ("C4", np.float64, 256, 10_000_000, 4, 3, 9.5e9),
                                                    ("C3", np.complex64, 512, 100_000_000, 8, 1, 24e9)])
def test_workspace_footprint_of_the_baseline_configurations(name, Z, n, Np, M, C, limit):
    """Plan-owned device memory of the BASELINE configurations with their point sets in place (C2 4.52 GB, C3 27.8 GB
    in round 5 against ~2.3 / ~11 GB of plan memory in the reference, which aliases the caller's points — src/plan.jl:37-60,
    src/blocking/gpu.jl:41-69): round 6 stopped creating the D-dimensional rocFFT plans (and their work buffer: a whole grid) on plans that
    run the library's own FFT passes, and keeps the compact spectrum only.  nufft_workspace_breakdown names every buffer and sums to
    nufft_info.workspace_bytes."""
    from nufft_pkg import nufft
    plan = nufft.PlanNUFFT(Z, (n, n, n), m=M, sigma=2.0, ntransforms=C, backend=nufft.ROCBackend(0))
    T = torch.float32 if np.dtype(Z) in (np.dtype(np.float32), np.dtype(np.complex64)) else torch.float64
    g = torch.Generator(device="cuda").manual_seed(1)
    xs = tuple(torch.rand(Np, dtype=T, device="cuda", generator=g) * (2 * np.pi) for _ in range(3))
    nufft.set_points(plan, xs)
    torch.cuda.synchronize()
    total = int(plan.info().workspace_bytes)
    parts = plan.workspace_breakdown()
    print(f"{name}: workspace {total / 1e9:.2f} GB: " + ", ".join(f"{k} {v / 1e9:.3f}" for k, v in sorted(parts.items(), key=lambda kv: -kv[1])))
    assert sum(parts.values()) == total
    assert {"us", "uhat", "sorted"} <= set(parts) and "rocfft_work" not in parts
    assert total <= limit, (name, total)
    # the transforms still work from this footprint (type 1 + type 2, finite and non-trivial)
    v = tuple(torch.randn(Np, dtype=plan.Z, device="cuda", generator=g) for _ in range(C))
    u = tuple(torch.empty(plan.shape, dtype=plan.eltype, device="cuda") for _ in range(C))
    nufft.exec_type1(u if C > 1 else u[0], plan, v if C > 1 else v[0])
    o = tuple(torch.empty(Np, dtype=plan.Z, device="cuda") for _ in range(C))
    nufft.exec_type2(o if C > 1 else o[0], plan, u if C > 1 else u[0])
    torch.cuda.synchronize()
    assert bool(torch.isfinite(torch.view_as_real(u[0])).all()) and float(u[0].abs().max()) > 0
    assert bool(torch.isfinite(o[0] if not o[0].is_complex() else torch.view_as_real(o[0])).all())
    assert int(plan.info().workspace_bytes) == total          # nothing allocated by the transforms
