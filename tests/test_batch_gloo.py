"""world_size-2 test of the plan-batch sharding logic on CPU (gloo).  The executor is the oracle (the
HIP path needs a GPU); what is exercised is the rank/shard assignment and the final gather."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import nufft_oracle as O

DIMS, NP, M, SIGMA, NPLANS = (12, 10, 8), 200, 4, 2.0, 5


class OracleExecutor:
    device = torch.device("cpu")

    def __init__(self):
        self.plan = O.OraclePlan(DIMS, is_real=True, M=M, sigma=SIGMA)

    def out_shape(self):
        return tuple(reversed(self.plan.size))

    def out_dtype(self):
        return torch.complex128

    def type1(self, points, values, out):
        O.set_points(self.plan, [x.numpy() for x in points])
        out.copy_(torch.from_numpy(O.exec_type1(self.plan, values.numpy())))
        return out

    def type2(self, points, uhat, out):
        O.set_points(self.plan, [x.numpy() for x in points])
        out.copy_(torch.from_numpy(O.exec_type2(self.plan, uhat.numpy())))
        return out


def _problem(b):
    rng = np.random.default_rng(100 + b)
    xs = tuple(torch.from_numpy(rng.random(NP) * O.TWO_PI) for _ in DIMS)
    v = torch.from_numpy(rng.standard_normal(NP))
    return xs, v


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nufft_pkg import nufft
    batch_mod = __import__("nonuniformffts_jl_amd.batch", fromlist=["PlanBatch"])
    batch = batch_mod.PlanBatch(NPLANS, OracleExecutor())
    assert batch.owned == list(range(rank, NPLANS, world))
    pts, vals = zip(*[_problem(b) for b in batch.owned])
    outs = batch.exec_type1(pts, vals)
    gathered = batch.gather_type1(outs, dst=0)
    if rank == 0:
        q.put([g.numpy() for g in gathered])
    else:
        assert gathered is None
    dist.barrier()
    dist.destroy_process_group()


def _worker_components(rank, world, port, q):
    """ntransforms = 3 sharded over two ranks: rank 0 owns components 0 and 2, rank 1 component 1; the same points everywhere."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nufft_pkg import nufft  # noqa: F401
    batch_mod = __import__("nonuniformffts_jl_amd.batch", fromlist=["PlanBatch"])
    C = 3
    batch = batch_mod.PlanBatch.from_ntransforms(C, OracleExecutor())
    assert batch.owned == list(range(rank, C, world)) and batch.shared_points
    xs, _ = _problem(0)
    vals = [_problem(10 + c)[1] for c in batch.owned]            # component c's value vector
    outs = batch.exec_components_type1(xs, vals)
    gathered = batch.gather_type1(outs, dst=0)
    if rank == 0:
        q.put([g.numpy() for g in gathered])
    else:
        assert gathered is None
    dist.barrier()
    dist.destroy_process_group()


def test_components_of_one_transform_shard_over_ranks():
    """`PlanBatch.from_ntransforms` (north_star: "independent transforms (ntransforms or batched plans) shard ... across the
    GPUs"): C = 3 components on 2 ranks, gathered in component order on rank 0, equal to the oracle's ntransforms = 3 plan."""
    world, C = 2, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_components, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        gathered = q.get(timeout=120)
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs)
    assert len(gathered) == C
    xs, _ = _problem(0)
    plan3 = O.OraclePlan(DIMS, is_real=True, M=M, sigma=SIGMA, ntransforms=C)
    O.set_points(plan3, [x.numpy() for x in xs])
    ref = O.exec_type1(plan3, [_problem(10 + c)[1].numpy() for c in range(C)])
    for c in range(C):
        assert np.allclose(gathered[c], np.asarray(ref[c]), rtol=1e-13, atol=1e-13)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_batch_shards_round_robin_and_gathers_in_batch_order():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        gathered = q.get(timeout=120)
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs)
    assert len(gathered) == NPLANS
    ex = OracleExecutor()
    for b in range(NPLANS):
        xs, v = _problem(b)
        ref = torch.empty(ex.out_shape(), dtype=torch.complex128)
        ex.type1(xs, v, ref)
        assert np.array_equal(gathered[b], ref.numpy())          # same code, same inputs: bit-exact


def test_owned_indices():
    from nufft_pkg import nufft  # noqa: F401
    batch_mod = __import__("nonuniformffts_jl_amd.batch", fromlist=["owned_indices"])
    assert batch_mod.owned_indices(8, 3, 8) == [3]
    assert batch_mod.owned_indices(8, 1, 2) == [1, 3, 5, 7]
    assert batch_mod.owned_indices(5, 1, 2) == [1, 3]
    assert sorted(sum((batch_mod.owned_indices(11, r, 4) for r in range(4)), [])) == list(range(11))
