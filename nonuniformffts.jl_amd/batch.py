"""Sharding of independent NUFFT plans — or of the components of one `ntransforms = C` plan — over the GPUs of one node
(BASELINE configs[4]; north_star: "independent transforms (ntransforms or batched plans) shard embarrassingly").

The reference has no multi-GPU code.  The path shards embarrassingly: independent
plans (or the components of a batch) share nothing, so plan ``b`` of a batch lives on rank
``b mod world_size`` — one process per GPU, no collective on the data path.  The only communication
is the optional *final gather* of the output spectra to one consumer rank (``gather_type1``), which
maps to one RCCL gather over xGMI (``torch.distributed`` backend ``"nccl"``) or gloo on CPU.

The executor is injectable so that the rank/shard logic can be exercised on CPU with the ``gloo``
backend (tests pass an oracle-backed executor); on a GPU box the default executor is the HIP plan.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import torch
import torch.distributed as dist


def owned_indices(nplans: int, rank: int, world_size: int) -> List[int]:
    """Indices of the batch that ``rank`` executes: b ≡ rank (mod world_size)."""
    return list(range(rank, nplans, world_size))


class HipExecutor:
    """Default executor: one ``PlanNUFFT`` on this rank's GPU, reused for every owned problem."""

    def __init__(self, Z, dims, device_index: int, **plan_kwargs):
        from . import plan as _plan
        self._p = _plan
        self.plan = _plan.PlanNUFFT(Z, dims, backend=_plan.ROCBackend(device_index), **plan_kwargs)
        self.device = self.plan.device

    def out_shape(self):
        return self.plan.shape

    def out_dtype(self):
        return self.plan.eltype

    def type1(self, points, values, out):
        self._p.set_points(self.plan, points)
        self._p.exec_type1(out, self.plan, values)
        return out

    def type2(self, points, uhat, out):
        self._p.set_points(self.plan, points)
        self._p.exec_type2(out, self.plan, uhat)
        return out

    # components of one transform: the points are set once, then one exec per owned component
    def type1_shared_points(self, points, values_list, outs):
        self._p.set_points(self.plan, points)
        for v, o in zip(values_list, outs):
            self._p.exec_type1(o, self.plan, v)
        return outs

    def type2_shared_points(self, points, uhats, outs):
        self._p.set_points(self.plan, points)
        for u, o in zip(uhats, outs):
            self._p.exec_type2(o, self.plan, u)
        return outs


class PlanBatch:
    """A batch of ``nplans`` independent transforms of identical shape, sharded one-per-rank.

    ``executor`` must provide ``type1(points, values, out)``, ``type2(points, uhat, out)``,
    ``out_shape()``, ``out_dtype()`` and a ``device`` attribute.
    """

    def __init__(self, nplans: int, executor, group=None):
        self.nplans = int(nplans)
        self.executor = executor
        self.group = group
        if dist.is_available() and dist.is_initialized():
            self.rank = dist.get_rank(group)
            self.world_size = dist.get_world_size(group)
        else:
            self.rank, self.world_size = 0, 1
        self.owned = owned_indices(self.nplans, self.rank, self.world_size)

    @classmethod
    def from_ntransforms(cls, ntransforms: int, executor, group=None) -> "PlanBatch":
        """The C components of ONE transform (``PlanNUFFT(...; ntransforms = Val(C))``, src/plan.jl:166-176) sharded over the
        ranks: component c lives on rank c mod world_size, every rank holds the same point set (`exec_components_type1`
        takes it once).  Components share nothing but the read-only points, so this is the same
        embarrassingly parallel split as independent plans; ``gather_type1`` returns the C spectra in component order.
        ``executor`` is a single-component executor (the rank's plan has ntransforms = 1 and is reused per component)."""
        b = cls(ntransforms, executor, group)
        b.shared_points = True
        return b

    shared_points = False

    def exec_components_type1(self, points, values_owned: Sequence) -> List[torch.Tensor]:
        """`from_ntransforms` batches: the owned components of the transform on the common point set; ``values_owned[i]`` is
        the value vector of component ``self.owned[i]``."""
        assert self.shared_points and len(values_owned) == len(self.owned)
        outs = [torch.empty(self.executor.out_shape(), dtype=self.executor.out_dtype(), device=self.executor.device) for _ in self.owned]
        if hasattr(self.executor, "type1_shared_points"):
            self.executor.type1_shared_points(points, list(values_owned), outs)
        else:
            for v, o in zip(values_owned, outs):
                self.executor.type1(points, v, o)
        return outs

    def exec_components_type2(self, points, uhats_owned: Sequence, outs: Sequence) -> List[torch.Tensor]:
        assert self.shared_points and len(uhats_owned) == len(outs) == len(self.owned)
        if hasattr(self.executor, "type2_shared_points"):
            self.executor.type2_shared_points(points, list(uhats_owned), list(outs))
        else:
            for u, o in zip(uhats_owned, outs):
                self.executor.type2(points, u, o)
        return list(outs)

    def exec_type1(self, points: Sequence, values: Sequence) -> List[torch.Tensor]:
        """Runs every owned problem; ``points[i]`` / ``values[i]`` belong to ``self.owned[i]``."""
        assert len(points) == len(values) == len(self.owned)
        outs = []
        for x, v in zip(points, values):
            out = torch.empty(self.executor.out_shape(), dtype=self.executor.out_dtype(), device=self.executor.device)
            outs.append(self.executor.type1(x, v, out))
        return outs

    def exec_type2(self, points: Sequence, uhats: Sequence, outs: Sequence) -> List[torch.Tensor]:
        assert len(points) == len(uhats) == len(outs) == len(self.owned)
        return [self.executor.type2(x, u, o) for x, u, o in zip(points, uhats, outs)]

    def gather_type1(self, local_outs: Sequence[torch.Tensor], dst: int = 0) -> Optional[List[torch.Tensor]]:
        """The single collective of the path: gathers all ``nplans`` spectra on rank ``dst`` in batch
        order.  Ranks own ⌈nplans / world⌉ or ⌊nplans / world⌋ problems; rounds with a missing problem
        send an empty placeholder that is dropped on the destination."""
        if self.world_size == 1:
            return list(local_outs)
        rounds = (self.nplans + self.world_size - 1) // self.world_size
        shape, dtype, device = self.executor.out_shape(), self.executor.out_dtype(), self.executor.device
        result: List[Optional[torch.Tensor]] = [None] * self.nplans
        for r in range(rounds):
            have = r < len(local_outs)
            send = local_outs[r].contiguous() if have else torch.zeros(shape, dtype=dtype, device=device)
            recv = [torch.empty(shape, dtype=dtype, device=device) for _ in range(self.world_size)] if self.rank == dst else None
            # complex spectra travel as their (re, im) real views: gloo has no complex support and the
            # bytes are identical
            cplx = send.is_complex()
            send_r = torch.view_as_real(send) if cplx else send
            recv_r = ([torch.view_as_real(t) for t in recv] if cplx else recv) if recv is not None else None
            dist.gather(send_r, recv_r, dst=dst, group=self.group)
            if self.rank == dst:
                for src in range(self.world_size):
                    b = r * self.world_size + src
                    if b < self.nplans:
                        result[b] = recv[src]
        return result if self.rank == dst else None
