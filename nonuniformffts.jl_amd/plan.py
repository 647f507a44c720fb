"""Host-side mirror of the reference's public interface for the GPU hot path:
``PlanNUFFT`` / ``set_points!`` / ``exec_type1!`` / ``exec_type2!``
(reference src/plan.jl:326-599, src/set_points.jl:33-88, src/NonuniformFFTs.jl:148-291).

Same names, argument meaning and error behaviour as the reference; Julia's ``!`` suffix is
dropped (``set_points``, ``exec_type1``, ``exec_type2``; ``*_`` aliases exist).  Every call goes
through the C ABI of ``libnufft_mi355x.so`` — PyTorch is used only for device memory and streams.

Array convention.  The reference is column-major: a Julia array of size ``(N1, N2, N3)`` has
dimension 1 fastest.  Here such an array is a C-contiguous ``torch`` tensor of shape
``(N3, N2, N1)`` (same bytes).  ``size(p)`` returns the Julia-order dims, ``p.shape`` the tensor
shape.

Error mapping: Julia ``ArgumentError`` -> ``ValueError``; ``DimensionMismatch`` ->
:class:`DimensionMismatch` (a ``ValueError`` subclass).
"""
from __future__ import annotations

import ctypes as C
import math
import os
from dataclasses import dataclass
from typing import Optional, Sequence, Tuple, Union

import torch

from . import _lib
from ._lib import lib

# NUFFT_* variables that belong to the Python / test harness itself, not to the library's development switches
_HARNESS_ENV = frozenset({"NUFFT_LIB_PATH", "NUFFT_BENCH_SHARE_GPU", "NUFFT_TEST_ERRLOG", "NUFFT_TEST_EXTRA_SEEDS"})


class DimensionMismatch(ValueError):
    """Julia ``DimensionMismatch`` (src/NonuniformFFTs.jl:92-114, src/blocking/gpu.jl:86)."""


# ---- small value types of the reference's API (src/Kernels/Kernels.jl:8-46) -------------------
#: spreading engines of the library (include/nufft_mi355x.h NUFFT_SPREAD_*)
_SPREAD_METHODS = {"auto": 0, "lds_tiles": 1, "mfma_patches": 2, "marching_ring": 3}


@dataclass(frozen=True)
class HalfSupport:
    M: int


@dataclass(frozen=True)
class Direct:
    """Kernels.Direct evaluation mode."""


@dataclass(frozen=True)
class FastApproximation:
    """Kernels.FastApproximation evaluation mode (piecewise polynomial)."""


@dataclass(frozen=True)
class BackwardsKaiserBesselKernel:
    """src/Kernels/kaiser_bessel_backwards.jl (default kernel); ``beta`` overrides the optimal β."""
    beta: Optional[float] = None


@dataclass(frozen=True)
class KaiserBesselKernel:
    """src/Kernels/kaiser_bessel.jl; ``beta`` overrides the optimal β."""
    beta: Optional[float] = None


@dataclass(frozen=True)
class GaussianKernel:
    """src/Kernels/gaussian.jl; ``ell`` = ℓ/Δx overrides the optimal width."""
    ell: Optional[float] = None


@dataclass(frozen=True)
class BSplineKernel:
    """src/Kernels/bspline.jl (order 2M, no shape parameter)."""


_KERNEL_IDS = {BackwardsKaiserBesselKernel: 0, KaiserBesselKernel: 1, GaussianKernel: 2, BSplineKernel: 3}


@dataclass(frozen=True)
class ROCBackend:
    """The backend value a ROC plan carries (KernelAbstractions ``ROCBackend()``)."""
    device: int = 0


class PointWeights:
    """``nonuniform = (v, n) -> v * weights[n]`` (src/plan.jl:112-124): a real device vector ``T[Np]``."""

    def __init__(self, weights: torch.Tensor):
        self.weights = weights


class ModeFactors:
    """``uniform = (w, idx) -> w * factors[idx]`` (src/plan.jl:126-146): a real device array with the
    dimensions of one uniform array (``size(p)``; tensor shape = reversed), shared by all components."""

    def __init__(self, factors: torch.Tensor):
        self.factors = factors


class NUFFTCallbacks:
    """``NUFFTCallbacks(; nonuniform, uniform)`` — src/plan.jl:146-164.  Arbitrary closures cannot cross the
    C ABI (SURVEY.md §7 "hard parts"); the two documented uses are offered as a fixed menu, fused into the
    spreading / interpolation and deconvolution kernels exactly where the reference calls the closures:
    ``nonuniform=PointWeights(w)`` and ``uniform=ModeFactors(f)``.  Anything else raises."""

    def __init__(self, nonuniform=None, uniform=None):
        if nonuniform is not None and not isinstance(nonuniform, PointWeights):
            raise NotImplementedError(
                "user callbacks are arbitrary closures and cannot cross the C ABI; use PointWeights(w) "
                "or apply the function outside the transform")
        if uniform is not None and not isinstance(uniform, ModeFactors):
            raise NotImplementedError(
                "user callbacks are arbitrary closures and cannot cross the C ABI; use ModeFactors(f) "
                "or apply the function outside the transform")
        self.nonuniform = nonuniform
        self.uniform = uniform

    def _struct(self, p: "PlanNUFFT", npoints: int):
        cb = _lib.NufftCallbacks()
        for name, obj, shape in (("point_weights", self.nonuniform and self.nonuniform.weights, (npoints,)),
                                 ("mode_factors", self.uniform and self.uniform.factors, p.shape)):
            if obj is None:
                continue
            if not isinstance(obj, torch.Tensor) or obj.dtype != p.T or obj.device != p.device or not obj.is_contiguous():
                raise ValueError(f"{name} must be a contiguous {p.T} tensor on {p.device}")
            if tuple(obj.shape) != tuple(shape):
                raise DimensionMismatch(f"{name}: expected tensor shape {tuple(shape)}, got {tuple(obj.shape)}")
            setattr(cb, name, obj.data_ptr())
        return cb


def transform_point_convention(x):
    """``_transform_point_convention`` (src/abstractNFFTs.jl:147-155) for host-side use; passing this
    function (or ``"nfft"``) as ``point_transform`` selects the device implementation of the same map."""
    twopi = 2 * math.pi
    x = -(twopi * x)
    return torch.where(x < 0, x + twopi, x) if isinstance(x, torch.Tensor) else (x + twopi if x < 0 else x)


def default_kernel(backend=None):
    """ext/NonuniformFFTsAMDGPUExt.jl:54."""
    return BackwardsKaiserBesselKernel()


def default_kernel_evalmode(backend=None):
    """ext/NonuniformFFTsAMDGPUExt.jl:56 (Direct on ROC)."""
    return Direct()


_REAL = {torch.float32: torch.float32, torch.float64: torch.float64,
         torch.complex64: torch.float32, torch.complex128: torch.float64}
_CPLX = {torch.float32: torch.complex64, torch.float64: torch.complex128}


def _to_torch_dtype(Z) -> torch.dtype:
    if isinstance(Z, torch.dtype):
        if Z not in _REAL:
            raise ValueError(f"unsupported element type {Z}")
        return Z
    if Z is float:
        return torch.float64
    if Z is complex:
        return torch.complex128
    try:
        import numpy as np
        return {np.dtype("float32"): torch.float32, np.dtype("float64"): torch.float64,
                np.dtype("complex64"): torch.complex64, np.dtype("complex128"): torch.complex128}[np.dtype(Z)]
    except Exception as exc:
        raise ValueError(f"unsupported element type {Z!r}") from exc


def _check(code: int):
    if code == _lib.OK:
        return
    msg = _lib.error_message(code)
    if code == _lib.ERR_DIM_MISMATCH:
        raise DimensionMismatch(msg)
    if code in (_lib.ERR_INVALID_ARG, _lib.ERR_SIZE_TOO_SMALL, _lib.ERR_LDS_TOO_SMALL, _lib.ERR_UNSUPPORTED,
                _lib.ERR_NO_POINTS, _lib.ERR_NO_DEVICE):
        raise ValueError(msg)
    raise RuntimeError(msg)


def _ptr_table(tensors: Sequence[torch.Tensor]):
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


class PlanNUFFT:
    """``PlanNUFFT([Z = ComplexF64], dims; m, σ, kernel, ntransforms, backend, kernel_evalmode,
    fftshift, gpu_method, ...)`` — reference src/plan.jl:166-325 (docs), :467-599 (constructors).

    Only GPU plans exist here (``backend=ROCBackend(device)``); ``gpu_method`` must be
    ``"shared_memory"`` (the LDS-tile path this package implements).  Pass ``backend=None`` for a
    host-only plan that exposes the plan-time parameter math without touching a GPU.
    """

    def __init__(self, Z=torch.complex128, dims: Union[int, Sequence[int]] = None, *,
                 m: Union[int, HalfSupport] = 4, sigma: float = 2.0, σ: Optional[float] = None,
                 kernel=None, ntransforms: int = 1, backend=ROCBackend(0),
                 kernel_evalmode=None, fftshift: bool = False, gpu_method: str = "shared_memory",
                 sort_points: bool = False, synchronise: bool = False, block_size=None, point_transform=None,
                 tile_dims: Optional[Sequence[int]] = None, interp_tile_dims: Optional[Sequence[int]] = None,
                 bin_log2: int = 0, lds_budget_bytes: int = 0, spread_threads: int = 0, interp_threads: int = 0,
                 spread_method: Union[str, int] = "auto",
                 kernel_param_dim: Optional[Sequence[float]] = None, oversampled_dims: Optional[Sequence[int]] = None,
                 options: Optional[dict] = None):
        if dims is None:           # PlanNUFFT(dims; ...) form: ComplexF64 by default (src/plan.jl:597-599)
            Z, dims = torch.complex128, Z
        if isinstance(dims, int):
            dims = (dims,)
        self.Z = _to_torch_dtype(Z)
        self.T = _REAL[self.Z]
        self.is_complex = self.Z.is_complex
        self.dims = tuple(int(n) for n in dims)
        if σ is not None:
            sigma = σ
        M = m.M if isinstance(m, HalfSupport) else int(m)
        kernel = default_kernel(backend) if kernel is None else kernel
        if isinstance(kernel, type):
            kernel = kernel()
        if type(kernel) not in _KERNEL_IDS:
            raise ValueError("kernel must be BackwardsKaiserBesselKernel, KaiserBesselKernel, GaussianKernel or BSplineKernel")
        kernel_evalmode = default_kernel_evalmode(backend) if kernel_evalmode is None else kernel_evalmode
        if isinstance(kernel_evalmode, type):
            kernel_evalmode = kernel_evalmode()
        if not isinstance(kernel_evalmode, (Direct, FastApproximation)):
            raise ValueError("kernel_evalmode must be Direct() or FastApproximation()")
        if gpu_method not in ("global_memory", "shared_memory"):
            raise ValueError("expected gpu_method ∈ (:global_memory, :shared_memory)")   # src/blocking/gpu.jl:26
        # gpu_method = :global_memory and sort_points = True() only change how the reference schedules the
        # same arithmetic (src/spreading/gpu.jl:168-186, src/blocking/gpu.jl:126-137); here every plan runs
        # the LDS-tile path on points that are always bin-sorted into plan-owned storage, so both are accepted.
        if point_transform in (None, "identity"):
            ptrans = 0
        elif point_transform in ("nfft", "abstractnffts") or point_transform is transform_point_convention:
            ptrans = 1                                              # src/abstractNFFTs.jl:147-155
        else:
            raise ValueError("point_transform must be None (identity) or the AbstractNFFTs convention: "
                             "arbitrary closures cannot cross the C ABI")
        self.point_transform = ptrans
        self.kernel = kernel
        self.kernel_evalmode = kernel_evalmode
        self.backend = backend
        self.fftshift = bool(fftshift)
        self.synchronise = bool(synchronise)
        self.gpu_method = gpu_method
        self._ntransforms = int(ntransforms)
        self._points = None
        self._handle = C.c_void_p()

        prm = _lib.NufftParams()
        prm.dtype = _lib.F32 if self.T == torch.float32 else _lib.F64
        prm.is_complex = int(self.is_complex)
        prm.ndim = len(self.dims)
        if not 1 <= prm.ndim <= 3:
            raise ValueError("only 1-, 2- and 3-dimensional transforms are supported")
        for d, n in enumerate(self.dims):
            prm.N[d] = n
        prm.half_support = M
        prm.sigma = float(sigma)
        prm.kernel = _KERNEL_IDS[type(kernel)]
        kparam = getattr(kernel, "beta", None) if not isinstance(kernel, GaussianKernel) else kernel.ell
        prm.kernel_param = 0.0 if kparam is None else float(kparam)
        prm.evalmode = _lib.EVAL_DIRECT if isinstance(kernel_evalmode, Direct) else _lib.EVAL_FAST_APPROXIMATION
        prm.ntransforms = self._ntransforms
        prm.fftshift = int(self.fftshift)
        prm.point_transform = ptrans
        prm.gpu_method = 0 if gpu_method == "shared_memory" else 1      # NUFFT_METHOD_*: scheduling-only, both accepted
        # what a binding that already holds the reference's plan forwards verbatim (julia/ext/NonuniformFFTsMI355XExt.jl):
        # p.kernels[d].β (σ / Δx for the Gaussian) and gridsize(p.kernels[d]) per dimension
        if kernel_param_dim is not None:
            for d, v in enumerate(kernel_param_dim):
                prm.kernel_param_dim[d] = float(v)
        if oversampled_dims is not None:
            for d, n in enumerate(oversampled_dims):
                prm.N_over[d] = int(n)
        if backend is None:
            prm.device = -1
            self.device = None
        else:
            dev = backend.device if isinstance(backend, ROCBackend) else torch.device(backend).index or 0
            if not torch.cuda.is_available():
                raise RuntimeError("no HIP device available: GPU plans need an MI355X (there is no CPU fallback)")
            torch.cuda.init()
            prm.device = int(dev)
            self.device = torch.device("cuda", int(dev))
        if tile_dims is not None:
            for d, n in enumerate(tile_dims):
                prm.tile_dims[d] = int(n)
        if interp_tile_dims is not None:
            for d, n in enumerate(interp_tile_dims):
                prm.interp_tile_dims[d] = int(n)
        prm.bin_log2 = int(bin_log2)
        prm.lds_budget_bytes = int(lds_budget_bytes)
        prm.spread_threads = int(spread_threads)
        prm.interp_threads = int(interp_threads)
        prm.spread_method = spread_method if isinstance(spread_method, int) else _SPREAD_METHODS[spread_method]
        prm.struct_size = C.sizeof(_lib.NufftParams)
        # development switches (nufft_params.options): the library reads no environment variable.  This module is the development
        # harness, so it forwards the NUFFT_* variables of its own process (helper scripts and tests keep their command lines);
        # explicit `options` win.  `plan.options` returns what the plan holds.
        opts = {k: v for k, v in os.environ.items() if k.startswith("NUFFT_") and k not in _HARNESS_ENV and v != ""}
        opts.update({str(k): str(v) for k, v in (options or {}).items()})
        self._options_text = ";".join(f"{k}={v}" for k, v in sorted(opts.items())).encode()      # (kept alive for the call)
        prm.options = self._options_text if opts else None
        _check(lib.nufft_plan_create_ex(C.byref(self._handle), C.byref(prm)))
        self._info = _lib.NufftInfo()
        _check(lib.nufft_plan_info(self._handle, C.byref(self._info)))
        if self.synchronise:
            _check(lib.nufft_set_timing(self._handle, 1))

    @property
    def options(self) -> str:
        """The development switches this plan was created with ("NUFFT_NAME=value;...", empty by default)."""
        return lib.nufft_plan_options(self._handle).decode()

    def workspace_breakdown(self) -> dict:
        """Plan-owned device bytes right now, buffer by buffer (sums to info().workspace_bytes)."""
        buf = C.create_string_buffer(4096)
        _check(lib.nufft_workspace_breakdown(self._handle, buf, len(buf)))
        out = {}
        for item in buf.value.decode().split(";"):
            if item:
                k, v = item.split("=")
                out[k] = int(v)
        return out

    # ---- lifetime ----------------------------------------------------------------------------
    def close(self):
        h = getattr(self, "_handle", None)
        if h is not None and h.value:
            lib.nufft_plan_destroy(h)
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- queries (src/plan.jl:360-435) ---------------------------------------------------------
    def info(self) -> _lib.NufftInfo:
        _check(lib.nufft_plan_info(self._handle, C.byref(self._info)))
        return self._info

    def sort_columns_used(self) -> bool:
        """True if the last set_points grouped the points by (column, layer of bins) only (plans with ``info().sort_column`` > 0,
        point sets both rings serve); reads two device flags back and synchronises."""
        out = C.c_int(0)
        _check(lib.nufft_sort_columns_used(self._handle, C.byref(out), self._stream()))
        return out.value == 1

    def sort_method_used(self) -> str:
        """Sort of the last set_points: "column_layers", "fine_bins" (histogram with global atomics) or "slabs" (fine bins in two
        levels through LDS: same array and offsets as "fine_bins"); reads device flags back and synchronises."""
        out = C.c_int(0)
        _check(lib.nufft_sort_columns_used(self._handle, C.byref(out), self._stream()))
        return {0: "fine_bins", 1: "column_layers", 2: "slabs"}[out.value]

    def spread_engine_used(self) -> str:
        """Engine that serves the point set of the last set_points: "lds_tiles" or "mfma_patches" (plans of the
        patch engine decide per point set on the device; this reads the decision back and synchronises)."""
        out = C.c_int(0)
        _check(lib.nufft_spread_engine_used(self._handle, C.byref(out), self._stream()))
        return {1: "lds_tiles", 2: "mfma_patches", 3: "marching_ring", 4: "marching_ring_dense"}[out.value]

    def interp_engine_used(self) -> str:
        """Engine that interpolates the point set of the last set_points in exec_type2: "lds_tiles" or "marching_ring"
        (decided on the device at set_points: the ring serves a point set while its heaviest task and its total work stay
        within the ring's measured advantage over the tile kernel); synchronises."""
        out = C.c_int(0)
        _check(lib.nufft_interp_engine_used(self._handle, C.byref(out), self._stream()))
        return {1: "lds_tiles", 2: "marching_ring"}[out.value]

    @property
    def size(self) -> Tuple[int, ...]:
        """size(p): dims of the uniform arrays in Julia order (N1÷2+1, N2, ...) for real Z."""
        return tuple(int(self._info.N_out[d]) for d in range(self.ndim))

    @property
    def shape(self) -> Tuple[int, ...]:
        """Shape of the corresponding C-contiguous torch tensor (reversed ``size``)."""
        return tuple(reversed(self.size))

    @property
    def ndim(self) -> int:
        return len(self.dims)

    @property
    def ntransforms(self) -> int:
        return self._ntransforms

    @property
    def eltype(self) -> torch.dtype:
        """eltype(p) = complex(Z) (src/plan.jl:360)."""
        return _CPLX[self.T]

    @property
    def oversampled_dims(self) -> Tuple[int, ...]:
        return tuple(int(self._info.N_over[d]) for d in range(self.ndim))

    @property
    def half_support(self) -> int:
        return int(self._info.half_support)

    @property
    def sigma(self) -> float:
        return float(self._info.sigma)

    σ = sigma

    @property
    def points(self):
        """p.points (src/plan.jl:412-413): the coordinate vectors given to set_points."""
        return self._points

    @property
    def num_points(self) -> int:
        return int(self.info().num_points)

    @property
    def timer(self) -> dict:
        """Per-stage milliseconds of the last run (TimerOutputs analogue).  As in the reference
        (src/plan.jl:397-403), GPU timings need ``synchronise=True``."""
        if not self.synchronise:
            import warnings
            warnings.warn("synchronisation is disabled on GPU: timings will be incorrect")
        buf = (C.c_float * _lib.NUM_STAGES)()
        _check(lib.nufft_get_stage_times(self._handle, buf))
        return {name: float(buf[i]) for i, name in enumerate(_lib.STAGE_NAMES) if buf[i] >= 0}

    def enable_timing(self, on: bool = True):
        self.synchronise = bool(on)
        _check(lib.nufft_set_timing(self._handle, int(on)))

    def __repr__(self):
        i = self._info
        D = self.ndim
        lines = [
            f"{D}-dimensional PlanNUFFT with input type {self.Z}:",
            f"  - backend: {self.backend}",
            f"  - kernel: {type(self.kernel).__name__}(shape parameter = {i.beta[0]}) with half-support M = {i.half_support}",
            f"  - kernel evaluation mode: {self.kernel_evalmode}",
            f"  - oversampling factor: σ = {i.sigma}",
            f"  - uniform dimensions: {self.size}",
            f"  - simultaneous transforms: {self.ntransforms}",
            f"  - frequency order: {'increasing' if self.fftshift else 'FFTW'} (fftshift = {self.fftshift})",
            f"  - block size: spreading {tuple(i.spread_tile[d] for d in range(D))} (interior only), interpolation "
            f"{tuple(i.interp_tile[d] for d in range(D))} (excluding 2M - 1 = {2 * i.half_support - 1} ghost cells in each direction), "
            f"sort bins {tuple(i.bin_dims[d] for d in range(D))}",
            f"  - GPU method: :{self.gpu_method} (LDS {i.lds_bytes_spread} B spread / {i.lds_bytes_interp} B interp)",
        ]
        return "\n".join(lines)

    # ---- plan-time host arrays ------------------------------------------------------------------
    def fourier_coefficients(self, dim: int):
        import numpy as np
        n = int(self._info.N_out[dim])
        out = np.empty(n, dtype=np.float64)
        _check(lib.nufft_plan_get_phi_hat(self._handle, dim, out.ctypes.data_as(C.POINTER(C.c_double)), n))
        return out

    def polynomial_coefficients(self, dim: int):
        import numpy as np
        shape = (int(self._info.npoly), 2 * int(self._info.half_support))
        out = np.empty(shape, dtype=np.float64)
        _check(lib.nufft_plan_get_poly_coefs(self._handle, dim, out.ctypes.data_as(C.POINTER(C.c_double)), out.size))
        return out

    def index_map(self, dim: int):
        import numpy as np
        n = int(self._info.N_out[dim])
        out = np.empty(n, dtype=np.int64)
        _check(lib.nufft_plan_get_index_map(self._handle, dim, out.ctypes.data_as(C.POINTER(C.c_int64)), n))
        return out

    # ---- internals -----------------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _require_gpu(self):
        if self.device is None:
            raise ValueError("host-only plan (backend=None) has no device path")

    def _check_nonuniform(self, vp: Sequence[torch.Tensor], what: str):
        """check_nufft_nonuniform_data, src/NonuniformFFTs.jl:105-114 + method signature :148-151."""
        if len(vp) != self.ntransforms:
            raise DimensionMismatch(f"wrong amount of data vectors (expected a tuple of {self.ntransforms} vectors)")
        if self._points is None:
            raise ValueError("set_points must be called before executing a transform")
        Np = self._points[0].numel()
        for v in vp:
            if not isinstance(v, torch.Tensor) or v.device != self.device:
                raise ValueError(f"{what} must be torch tensors on {self.device}")
            if v.dtype != self.Z:
                raise ValueError(f"non-uniform data must have element type {self.Z} (got {v.dtype})")
            if v.dim() != 1 or not v.is_contiguous():
                raise ValueError(f"{what} must be contiguous vectors")
            if v.numel() != Np:
                raise DimensionMismatch(
                    f"wrong length of data vector (it should match the number of points {Np}, got length {v.numel()})")

    def _check_uniform(self, us: Sequence[torch.Tensor]):
        """check_nufft_uniform_data, src/NonuniformFFTs.jl:92-103 + eltype check :154."""
        if len(us) != self.ntransforms:
            raise DimensionMismatch(f"wrong amount of arrays (expected a tuple of {self.ntransforms} arrays)")
        for u in us:
            if not isinstance(u, torch.Tensor) or u.device != self.device:
                raise ValueError(f"uniform data must be torch tensors on {self.device}")
            if u.dtype != self.eltype:
                raise ValueError(
                    f"uniform data must have the same accuracy as the created plan (got {u.dtype} values for a {self.Z} plan)")
            if u.dim() != self.ndim:
                raise DimensionMismatch(f"wrong dimensions of array (expected {self.ndim}-dimensional array)")
            if tuple(u.shape) != self.shape:
                raise DimensionMismatch(f"wrong dimensions of array (expected dimensions {self.size}, i.e. tensor shape {self.shape})")
            if not u.is_contiguous():
                raise ValueError("uniform data must be contiguous")


# ---- the three calls of the hot path -----------------------------------------------------------------

def set_points(p: PlanNUFFT, xp) -> PlanNUFFT:
    """``set_points!(p, points)`` — src/set_points.jl:33-88.

    ``points``: tuple of D device vectors (preferred); a single vector for 1-D plans; or an
    ``(Np, D)`` tensor (the memory layout of the reference's ``(d, Np)`` matrix), which is copied.
    """
    p._require_gpu()
    if isinstance(xp, torch.Tensor):
        if xp.dim() == 1:
            if p.ndim != 1:
                raise DimensionMismatch(f"expected {p.ndim}-dimensional points")
            xp = (xp,)
        elif xp.dim() == 2:
            if xp.shape[1] != p.ndim:
                raise DimensionMismatch(f"expected input matrix to have dimensions ({p.ndim}, Np)")
            xp = tuple(xp[:, d].contiguous() for d in range(p.ndim))
        else:
            raise ValueError("unexpected point container")
    xp = tuple(xp)
    if len(xp) != p.ndim:
        raise DimensionMismatch(f"expected {p.ndim}-dimensional points")
    for x in xp:
        if not isinstance(x, torch.Tensor):
            raise ValueError("unexpected point container: expected torch tensors")
        if x.dtype != p.T:
            raise ValueError(
                f"input points must have the same accuracy as the created plan (got {x.dtype} points for a {p.Z} plan)")
        if x.device != p.device:
            raise ValueError(f"unexpected point container: expected tensors on {p.device}, got {x.device}")
        if x.dim() != 1 or not x.is_contiguous():
            raise ValueError("unexpected point container: expected contiguous vectors")
    Np = xp[0].numel()
    if any(x.numel() != Np for x in xp):
        raise DimensionMismatch("input points must have the same length along all dimensions")
    tbl = _ptr_table(xp)
    _check(lib.nufft_set_points(p._handle, Np, tbl, p._stream()))
    p._points = xp
    return p


def exec_type1(us, p: PlanNUFFT, vp, *, callbacks: Optional[NUFFTCallbacks] = None):
    """``exec_type1!(ûs, p, vp)`` — src/NonuniformFFTs.jl:148-195.  Returns ``ûs``."""
    p._require_gpu()
    single = isinstance(us, torch.Tensor)
    us_t = (us,) if single else tuple(us)
    vp_t = (vp,) if isinstance(vp, torch.Tensor) else tuple(vp)
    p._check_uniform(us_t)
    p._check_nonuniform(vp_t, "input values")
    if callbacks is not None and (callbacks.nonuniform is not None or callbacks.uniform is not None):
        cb = callbacks._struct(p, vp_t[0].numel())
        _check(lib.nufft_exec_type1_cb(p._handle, _ptr_table(us_t), _ptr_table(vp_t), C.byref(cb), p._stream()))
    else:
        _check(lib.nufft_exec_type1(p._handle, _ptr_table(us_t), _ptr_table(vp_t), p._stream()))
    return us


def exec_type2(vp, p: PlanNUFFT, us, *, callbacks: Optional[NUFFTCallbacks] = None):
    """``exec_type2!(vp, p, ûs)`` — src/NonuniformFFTs.jl:237-291.  Returns ``vp``."""
    p._require_gpu()
    single = isinstance(vp, torch.Tensor)
    vp_t = (vp,) if single else tuple(vp)
    us_t = (us,) if isinstance(us, torch.Tensor) else tuple(us)
    p._check_uniform(us_t)
    p._check_nonuniform(vp_t, "output values")
    if callbacks is not None and (callbacks.nonuniform is not None or callbacks.uniform is not None):
        cb = callbacks._struct(p, vp_t[0].numel())
        _check(lib.nufft_exec_type2_cb(p._handle, _ptr_table(vp_t), _ptr_table(us_t), C.byref(cb), p._stream()))
    else:
        _check(lib.nufft_exec_type2(p._handle, _ptr_table(vp_t), _ptr_table(us_t), p._stream()))
    return vp


set_points_ = set_points
exec_type1_ = exec_type1
exec_type2_ = exec_type2


# ---- stage-level access (the backend-dispatched generic functions; used by tests and bench) ----------

def spread_from_points(p: PlanNUFFT, vp, zero: bool = True):
    """fill_with_zeros + spread_from_points!(::GPU, ...) (src/spreading/gpu.jl:134-214)."""
    vp_t = (vp,) if isinstance(vp, torch.Tensor) else tuple(vp)
    p._check_nonuniform(vp_t, "input values")
    if zero:
        _check(lib.nufft_fill_zeros(p._handle, p._stream()))
    _check(lib.nufft_spread(p._handle, _ptr_table(vp_t), p._stream()))


def interpolate(p: PlanNUFFT, vp):
    """interpolate!(::GPU, ...) from the plan's oversampled grids (src/interpolation/gpu.jl:40-118)."""
    vp_t = (vp,) if isinstance(vp, torch.Tensor) else tuple(vp)
    p._check_nonuniform(vp_t, "output values")
    _check(lib.nufft_interpolate(p._handle, _ptr_table(vp_t), p._stream()))
    return vp


def oversampled_grid(p: PlanNUFFT, component: int = 0, spectrum: bool = False) -> torch.Tensor:
    """Copy of p.data.us[c] (or ûs[c] for real plans) as a tensor with reversed axes."""
    p._require_gpu()
    Nover = p.oversampled_dims
    if spectrum:
        if p.is_complex:
            raise ValueError("complex plans transform in place: the spectrum lives in us")
        dims = (Nover[0] // 2 + 1,) + Nover[1:]
        out = torch.empty(tuple(reversed(dims)), dtype=p.eltype, device=p.device)
    else:
        out = torch.empty(tuple(reversed(Nover)), dtype=p.Z, device=p.device)
    _check(lib.nufft_copy_grid(p._handle, int(spectrum), component, C.c_void_p(out.data_ptr()),
                               out.numel() * out.element_size(), p._stream()))
    return out


def sort_result(p: PlanNUFFT):
    """(pointperm, cumulative number of points per sort bin) as numpy arrays (0-based)."""
    import numpy as np
    p._require_gpu()
    info = p.info()
    Np = int(info.num_points)
    nt = 1
    for d in range(p.ndim):
        nt *= int(info.nbins[d])
    perm = np.empty(max(Np, 1), dtype=np.int32)
    offs = np.empty(nt + 1, dtype=np.uint32)
    _check(lib.nufft_get_sort_result(p._handle, perm.ctypes.data_as(C.POINTER(C.c_int32)), perm.size,
                                     offs.ctypes.data_as(C.POINTER(C.c_uint32)), offs.size, p._stream()))
    return perm[:Np], offs
