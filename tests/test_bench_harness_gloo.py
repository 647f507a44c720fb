"""world_size-2 test (gloo, CPU) of bench.py's own multi-rank logic — the functions the N > 1 benchmark runs with RCCL:
`timed_steps` (warm-up, barrier-bracketed timed region, MAX over ranks) and `gather_components` (the single collective of
the path: every component's spectrum of every rank to rank 0).  The executor is a CPU stub (the HIP path needs a GPU);
what is exercised is the rank logic itself."""
import os
import socket
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

C, SHAPE, K, W = 3, (4, 6, 5), 4, 2


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    dev = torch.device("cpu")
    calls = {"warm": 0, "timed": 0}
    outs = tuple(torch.full(SHAPE, complex(rank + 1, 10 * (c + 1)), dtype=torch.complex128) for c in range(C))
    recv = [[torch.empty(SHAPE + (2,), dtype=torch.float64) for _ in range(world)] for _ in range(C)] if rank == 0 else None

    def step(k, events):
        calls["timed" if events is not None else "warm"] += 1
        time.sleep(0.02 * (1 + 3 * rank))          # rank 1 is four times slower: the job's time is the slowest rank's
        if events is not None:
            events.append(k)
            bench.gather_components(outs, recv, rank)

    dt, events = bench.timed_steps(step, K, W, dev, True)
    assert calls == {"warm": W, "timed": K} and events == list(range(K))
    result = {"rank": rank, "dt": dt}
    if rank == 0:
        # plain numpy through the queue: a torch tensor travels as a shared-memory handle that the parent may try to open
        # after this process has exited (observed: EOFError once in three runs)
        result["recv"] = [[t.numpy().copy() for t in row] for row in recv]
    q.put(result)
    dist.barrier()
    dist.destroy_process_group()


def test_timed_region_and_gather_with_two_ranks():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in procs:
        r = q.get(timeout=120)
        res[r["rank"]] = r
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # MAX over ranks: both ranks report the same time, and it is at least the slow rank's K steps
    assert abs(res[0]["dt"] - res[1]["dt"]) < 1e-9
    assert res[0]["dt"] >= K * 0.08 * 0.9
    # rank 0 holds every component of every rank
    for c in range(C):
        for src in range(2):
            got = torch.view_as_complex(torch.from_numpy(res[0]["recv"][c][src]))
            assert torch.all(got == complex(src + 1, 10 * (c + 1)))
