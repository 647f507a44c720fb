"""nonuniformffts.jl_amd — MI355X-native backend for the GPU hot path of NonuniformFFTs.jl.

Python host mirror of the reference interface (``PlanNUFFT``, ``set_points!``, ``exec_type1!``,
``exec_type2!``) above the C ABI of ``libnufft_mi355x.so`` (include/nufft_mi355x.h).  Importing
this package loads the HIP library and raises if it has not been built: there is no fallback.

The directory name contains a dot, so load it with ``nufft_pkg.py`` at the repo root
(``from nufft_pkg import nufft``) or ``importlib``.
"""
from ._lib import LIB_PATH, lib  # noqa: F401  (fails loudly if the extension is missing)
from .plan import (  # noqa: F401
    BackwardsKaiserBesselKernel, BSplineKernel, DimensionMismatch, Direct, FastApproximation, GaussianKernel,
    HalfSupport, KaiserBesselKernel, ModeFactors, NUFFTCallbacks, PlanNUFFT, PointWeights, ROCBackend, default_kernel,
    default_kernel_evalmode, exec_type1, exec_type1_, exec_type2, exec_type2_, interpolate, oversampled_grid,
    set_points, set_points_, sort_result, spread_from_points, transform_point_convention,
)
from .nfft_interface import NFFTPlan, plan_nfft  # noqa: F401

__all__ = [
    "PlanNUFFT", "NUFFTCallbacks", "PointWeights", "ModeFactors", "HalfSupport", "Direct", "FastApproximation",
    "BackwardsKaiserBesselKernel", "KaiserBesselKernel", "GaussianKernel", "BSplineKernel", "ROCBackend", "DimensionMismatch",
    "set_points", "exec_type1", "exec_type2", "set_points_", "exec_type1_", "exec_type2_",
    "NFFTPlan", "plan_nfft",
]
