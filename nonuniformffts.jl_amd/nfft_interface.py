"""AbstractNFFTs-compatible plan — mirror of ``NonuniformFFTs.NFFTPlan`` (src/abstractNFFTs.jl:52-229).

Differences from :class:`PlanNUFFT`, exactly as in the reference (:58-66): points live in [-1/2, 1/2),
the opposite Fourier sign convention is used (handled on the device by ``point_transform``,
``_transform_point_convention`` :147-155), uniform data is in increasing-frequency order
(``fftshift = true`` by default, :203) and only complex non-uniform data is supported.

    plan = NFFTPlan(xp, Ns)            # xp: (Np, D) tensor — the memory of Julia's (D, Np) matrix
    us = plan.adjoint_mul(vp)          # adjoint(p) * vp   : type-1, f̂_k = Σ_j f_j e^{+2πi k·x_j}
    wp = plan.mul(us)                  # p * us            : type-2, f_j = Σ_k f̂_k e^{-2πi k·x_j}
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import torch

from .plan import (BackwardsKaiserBesselKernel, BSplineKernel, GaussianKernel, KaiserBesselKernel, PlanNUFFT,
                   ROCBackend, default_kernel, exec_type1, exec_type2, set_points)


def convert_window_function(w, backend=None):
    """src/abstractNFFTs.jl:170-186 (NFFT.jl's :kaiser_bessel_rev is this package's KaiserBesselKernel)."""
    if not isinstance(w, str):
        return w
    return {"gauss": GaussianKernel(), "spline": BSplineKernel(), "kaiser_bessel_rev": KaiserBesselKernel(),
            "kaiser_bessel": BackwardsKaiserBesselKernel()}.get(w, default_kernel(backend))


def accuracy_params(m: Optional[int] = None, sigma: Optional[float] = None, reltol: Optional[float] = None):
    """``AbstractNFFTs.accuracyParams`` (third-party, not under /root/reference; NFFT.jl's published rule):
    reltol -> window width w = ceil(log10(1 / reltol)) + 1, m = (w - 1) ÷ 2 rounded up, σ = 2; explicit
    ``m`` / ``σ`` take precedence."""
    if m is None:
        if reltol is None:
            reltol = 1e-9
        w = math.ceil(math.log10(1.0 / reltol)) + 1
        m = (w + 1) // 2
    return int(m), (2.0 if sigma is None else float(sigma))


class NFFTPlan:
    def __init__(self, xp: torch.Tensor, Ns: Sequence[int], *, m: Optional[int] = None, sigma: Optional[float] = None,
                 σ: Optional[float] = None, reltol: Optional[float] = None, window=None, fftshift: bool = True,
                 sortNodes: bool = False, blocking: bool = True, precompute=None, **plan_kwargs):
        if not isinstance(xp, torch.Tensor) or not xp.is_floating_point():
            raise ValueError("xp must be a real floating-point tensor of shape (Np, D)")
        Ns = (int(Ns),) if isinstance(Ns, int) else tuple(int(n) for n in Ns)
        if xp.dim() == 1:
            xp = xp[:, None]
        if xp.dim() != 2 or xp.shape[1] != len(Ns):
            raise ValueError(f"expected input matrix to have dimensions ({len(Ns)}, Np)")
        T = xp.dtype
        Z = torch.complex64 if T == torch.float32 else torch.complex128
        m_actual, sigma_actual = accuracy_params(m, σ if σ is not None else sigma, reltol)
        backend = ROCBackend(xp.device.index or 0) if xp.is_cuda else None
        kernel = convert_window_function(window, backend) if window is not None else default_kernel(backend)
        self.p = PlanNUFFT(Z, Ns, m=m_actual, sigma=sigma_actual, kernel=kernel, backend=backend, fftshift=fftshift,
                           sort_points=bool(sortNodes), point_transform="nfft", **plan_kwargs)
        self.T = T
        self._np = 0
        if backend is not None:
            self.nodes(xp)

    # AbstractNFFTs.nodes!(p, xp), src/abstractNFFTs.jl:160-168
    def nodes(self, xp: torch.Tensor) -> "NFFTPlan":
        set_points(self.p, xp if xp.dim() == 2 else xp[:, None])
        self._np = int(xp.shape[0])
        return self

    @property
    def size_in(self):          # AbstractNFFTs.size_in, :128
        return self.p.size

    @property
    def size_out(self):         # AbstractNFFTs.size_out, :129
        return (self._np,)

    # mul!(vp, p, ûs), :132-137 — uniform to non-uniform
    def mul(self, us: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        if out is None:
            out = torch.empty(self._np, dtype=self.p.Z, device=us.device)
        exec_type2(out, self.p, us)
        return out

    # mul!(ûs, adjoint(p), vp), :139-145 — non-uniform to uniform
    def adjoint_mul(self, vp: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        if out is None:
            out = torch.empty(self.p.shape, dtype=self.p.eltype, device=vp.device)
        exec_type1(out, self.p, vp)
        return out

    def __repr__(self):
        return f"NonuniformFFTs.NFFTPlan{{{self.T}, {self.p.ndim}}} wrapping a PlanNUFFT:\n{self.p!r}"


def plan_nfft(xp: torch.Tensor, Ns, **kw) -> NFFTPlan:
    """``AbstractNFFTs.plan_nfft(xp, Ns; kw...)`` with this package as the backend (:28-50)."""
    return NFFTPlan(xp, Ns, **kw)
