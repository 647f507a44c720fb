"""Eager launches vs hipGraph replay of set_points + exec_type1 + exec_type2 (development helper).

Small transforms are launch-bound (a dozen short kernels); the whole call sequence only enqueues work on the stream it
is given (no allocation once the plan has seen Np points, no host read-back), so it can be captured once and replayed.

usage: python scripts/graph_probe.py [--reps 200]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nufft_pkg import nufft  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=200)
ap.add_argument("--cases", default="")
a = ap.parse_args()

CASES = [  # (label, Z, dims, Np)
    ("C1 1-D N=256 Np=1e4 f64", torch.float64, (256,), 10_000),
    ("2-D 128^2 Np=1e5 f64", torch.float64, (128, 128), 100_000),
    ("3-D 32^3 Np=1e4 f64", torch.float64, (32, 32, 32), 10_000),
    ("3-D 64^3 Np=1e5 f64", torch.float64, (64, 64, 64), 100_000),
    ("3-D 64^3 Np=1e5 c64", torch.complex64, (64, 64, 64), 100_000),
    ("3-D 128^3 Np=1e6 f64", torch.float64, (128, 128, 128), 1_000_000),
]

sel = [int(c) for c in a.cases.split(",")] if a.cases else range(len(CASES))
for label, Z, dims, Np in [CASES[i] for i in sel]:
    T = torch.float32 if Z in (torch.float32, torch.complex64) else torch.float64
    plan = nufft.PlanNUFFT(Z, dims, m=4, sigma=2.0, backend=nufft.ROCBackend(0))
    g = torch.Generator(device="cuda").manual_seed(1)
    xs = tuple(torch.rand(Np, dtype=T, device="cuda", generator=g) * (2 * np.pi) for _ in dims)
    v = torch.randn(Np, dtype=Z, device="cuda", generator=g)
    u = torch.empty(plan.shape, dtype=plan.eltype, device="cuda")
    w = torch.empty(Np, dtype=Z, device="cuda")

    def step():
        nufft.set_points(plan, xs)
        nufft.exec_type1(u, plan, v)
        nufft.exec_type2(w, plan, u)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        step()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / a.reps * 1e6
    u_ref, w_ref = u.clone(), w.clone()

    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    with torch.cuda.graph(graph):
        step()
    u.zero_(); w.zero_()
    graph.replay()
    torch.cuda.synchronize()
    ok = torch.equal(w, w_ref) or bool((w - w_ref).abs().max() <= 1e-6 * w_ref.abs().max())
    t0 = time.perf_counter()
    for _ in range(a.reps):
        graph.replay()
    torch.cuda.synchronize()
    rep = (time.perf_counter() - t0) / a.reps * 1e6
    print(f"{label:28s} eager {eager:8.1f} us   graph {rep:8.1f} us   x{eager / rep:.2f}   replay matches: {ok}", flush=True)
    plan.close()
