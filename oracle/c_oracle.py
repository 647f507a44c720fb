"""ctypes wrapper of oracle/libnufft_oracle.so (plain-C blocked CPU spreading / interpolation) and a
full type-1 / type-2 pipeline built from it + scipy's pocketfft.  TEST INFRASTRUCTURE ONLY — see the
header of oracle/nufft_oracle.py.  Used by tests (cross-check of the numpy oracle, larger parity
cases) and by bench.py's ``cpu_baseline`` leg ("port": the reference's Julia CPU backend cannot run
here).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import nufft_oracle as O

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libnufft_oracle.so")
_lib = None


def available() -> bool:
    return os.path.exists(_LIB)


def lib():
    global _lib
    if _lib is None:
        if not available():
            raise ImportError(f"{_LIB} not built: run `make -C oracle`")
        _lib = C.CDLL(_LIB)
        for name in ("oracle_spread_blocked", "oracle_interp_blocked", "oracle_deconv_truncate"):
            getattr(_lib, name).restype = C.c_int
        _lib.oracle_zero.restype = None
        _lib.oracle_num_threads.restype = C.c_int
    return _lib


def available_cpus() -> int:
    """CPUs this process may actually use: the scheduler affinity, capped by the cgroup CPU quota (a
    container that sees 256 logical CPUs may be limited to 16 CPUs' worth of time; more threads than that
    only add contention)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
            break
        except Exception:
            continue
    return n


_threads_set = False


def num_threads() -> int:
    """Threads the C oracle runs with: OMP_NUM_THREADS if the caller set it, else available_cpus()."""
    global _threads_set
    if not _threads_set:
        if "OMP_NUM_THREADS" not in os.environ:
            lib().oracle_set_num_threads(available_cpus())
        _threads_set = True
    return int(lib().oracle_num_threads())


def _ptrs(arrs):
    tbl = (C.c_void_p * len(arrs))()
    for i, a in enumerate(arrs):
        tbl[i] = a.ctypes.data
    return tbl


def _common(plan: O.OraclePlan):
    assert np.dtype(plan.dtype) == np.float64, "the C oracle is Float64 only"
    num_threads()                                   # applies the CPU quota once
    # Float32 points of a plan with coord_dtype = float32 are located with Float32 arithmetic (see nufft_oracle.c)
    lib().oracle_set_coord_f32(1 if (plan.coord_dtype is not None and np.dtype(plan.coord_dtype) == np.float32) else 0)
    D = plan.ndim
    N = (C.c_int64 * 3)(*(list(plan.Nover) + [1] * (3 - D)))
    coefs = np.ascontiguousarray(np.stack([plan.coefs[d] for d in range(D)]))   # [D][npoly][2M]
    betas = (C.c_double * 3)(*(list(plan.betas) + [0.0] * (3 - D)))
    xs = [np.ascontiguousarray(x, dtype=np.float64) for x in plan.points]
    return D, N, coefs, betas, xs


def _work(plan: O.OraclePlan, name: str, shape, dtype):
    """A plan-owned work array (the reference's plan owns its oversampled arrays `us` / `ûs`, src/plan.jl:33-34; allocating them anew
    in every transform would charge the first-touch page faults of a gigabyte to the transform)."""
    cache = plan.__dict__.setdefault("_c_work", {})
    a = cache.get(name)
    if a is None or a.shape != tuple(shape) or a.dtype != np.dtype(dtype):
        a = cache[name] = np.empty(shape, dtype=dtype)
    return a


def spread(plan: O.OraclePlan, vps, reuse: bool = False):
    """C restatement of spread_from_points!(::CPU, ..., ::BlockDataCPU, ...) (src/spreading/cpu_blocked.jl:94-168).
    Returns grids with reversed axes, like nufft_oracle.spread.  reuse: the grids are the plan's own work arrays (zeroed by a
    threaded fill), valid until the next call."""
    D, N, coefs, betas, xs = _common(plan)
    ncomp = 1 if plan.is_real else 2
    Np = len(xs[0])
    vs = [np.ascontiguousarray(v, dtype=np.float64 if plan.is_real else np.complex128) for v in vps]
    shape = tuple(reversed(plan.Nover))
    if reuse:
        us = [_work(plan, f"us{c}", shape, np.float64 if plan.is_real else np.complex128) for c in range(len(vs))]
        for u in us:
            lib().oracle_zero(C.c_void_p(u.ctypes.data), C.c_int64(u.size * ncomp))
    else:
        us = [np.zeros(shape, dtype=np.float64 if plan.is_real else np.complex128) for _ in vs]
    rc = lib().oracle_spread_blocked(C.c_int(D), N, C.c_int(plan.M), C.c_int(plan.evalmode), C.c_int(ncomp),
                                     C.c_void_p(coefs.ctypes.data), betas, C.c_int64(Np), _ptrs(xs),
                                     C.c_int(len(vs)), _ptrs(vs), _ptrs(us))
    assert rc == 0
    return us


def interpolate(plan: O.OraclePlan, us):
    """C restatement of interpolate!(::CPU, ..., ::BlockDataCPU, ...) (src/interpolation/cpu_blocked.jl:95-153)."""
    D, N, coefs, betas, xs = _common(plan)
    ncomp = 1 if plan.is_real else 2
    Np = len(xs[0])
    gs = [np.ascontiguousarray(u, dtype=np.float64 if plan.is_real else np.complex128) for u in us]
    vs = [np.empty(Np, dtype=np.float64 if plan.is_real else np.complex128) for _ in gs]
    rc = lib().oracle_interp_blocked(C.c_int(D), N, C.c_int(plan.M), C.c_int(plan.evalmode), C.c_int(ncomp),
                                     C.c_void_p(coefs.ctypes.data), betas, C.c_int64(Np), _ptrs(xs),
                                     C.c_int(len(gs)), _ptrs(gs), _ptrs(vs))
    assert rc == 0
    return vs


def _fft_workers():
    return num_threads()


def exec_type1(plan: O.OraclePlan, vp):
    """exec_type1! with the C spreading stage and scipy (pocketfft, all cores) for the FFT."""
    import scipy.fft as sfft
    single = not isinstance(vp, (list, tuple))
    us = spread(plan, [vp] if single else list(vp), reuse=True)
    norm = float(np.prod([O.TWO_PI / n for n in plan.Nover]))
    D = plan.ndim
    idx = [np.ascontiguousarray(plan.index_map[d], dtype=np.int64) for d in range(D)]
    inv = [np.ascontiguousarray(1.0 / np.asarray(plan.phihat[d], dtype=np.float64)) for d in range(D)]
    no = (C.c_int64 * 3)(*([len(i) for i in idx] + [1] * (3 - D)))
    outs = []
    for u in us:
        uh = sfft.rfftn(u, workers=_fft_workers()) if plan.is_real else sfft.fftn(u, workers=_fft_workers())
        uh = np.ascontiguousarray(uh, dtype=np.complex128)
        ns = (C.c_int64 * 3)(*(list(reversed(uh.shape)) + [1] * (3 - D)))
        out = np.empty(tuple(len(i) for i in reversed(idx)), dtype=np.complex128)
        # truncation + deconvolution + normalisation in one threaded C pass (checked against the numpy expression
        # uh[_gather_index] * norm / _deconv_factor by tests/test_c_oracle.py)
        rc = lib().oracle_deconv_truncate(C.c_int(D), ns, no, _ptrs(idx), _ptrs(inv), C.c_double(norm), C.c_void_p(uh.ctypes.data),
                                          C.c_void_p(out.ctypes.data))
        assert rc == 0
        outs.append(out.astype(plan.cdtype, copy=False))
    return outs[0] if single else outs


def exec_type2(plan: O.OraclePlan, uhat):
    import scipy.fft as sfft
    single = not isinstance(uhat, (list, tuple))
    fac = 1.0 / O._deconv_factor(plan)
    grids = []
    for w in ([uhat] if single else list(uhat)):
        if plan.is_real:
            shape = tuple(reversed((plan.Nover[0] // 2 + 1,) + plan.Nover[1:]))
        else:
            shape = tuple(reversed(plan.Nover))
        uh = np.zeros(shape, dtype=np.complex128)
        uh[O._gather_index(plan)] = np.asarray(w, dtype=np.complex128) * fac
        ntot = float(np.prod(plan.Nover))
        axes = tuple(range(plan.ndim))
        if plan.is_real:
            u = sfft.irfftn(uh, s=tuple(reversed(plan.Nover)), axes=axes, workers=_fft_workers()) * ntot
        else:
            u = sfft.ifftn(uh, workers=_fft_workers()) * ntot
        grids.append(u)
    vs = interpolate(plan, grids)
    return vs[0] if single else vs
