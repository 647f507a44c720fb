"""Quick per-stage timing of one configuration on cuda:0 (development helper, not the benchmark).

usage: python scripts/perf_probe.py [--n 256] [--np 10000000] [--m 4] [--sigma 2] [--z f64|c64|c128|f32]
                                    [--tile a,b,c] [--threads T] [--mode direct|poly] [--c 1] [--reps 5]
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
ap.add_argument("--n", type=int, default=256)
ap.add_argument("--dim", type=int, default=3)
ap.add_argument("--np", type=float, default=1e7)
ap.add_argument("--m", type=int, default=4)
ap.add_argument("--sigma", type=float, default=2.0)
ap.add_argument("--z", default="f64")
ap.add_argument("--tile", default="")
ap.add_argument("--itile", default="")
ap.add_argument("--bin", type=int, default=0)
ap.add_argument("--threads", type=int, default=0)
ap.add_argument("--lds", type=int, default=0)
ap.add_argument("--mode", default="direct")
ap.add_argument("--c", type=int, default=1)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--dist", default="uniform")
ap.add_argument("--kernel", default="bkb", choices=["bkb", "kb", "gaussian", "bspline"])
a = ap.parse_args()

Z = {"f64": torch.float64, "f32": torch.float32, "c64": torch.complex64, "c128": torch.complex128}[a.z]
T = torch.float32 if Z in (torch.float32, torch.complex64) else torch.float64
Np = int(a.np)
dims = (a.n,) * a.dim
kw = {}
if a.tile:
    kw["tile_dims"] = tuple(int(t) for t in a.tile.split(","))
if a.itile:
    kw["interp_tile_dims"] = tuple(int(t) for t in a.itile.split(","))
if a.bin:
    kw["bin_log2"] = a.bin
if a.threads:
    kw["spread_threads"] = a.threads
    kw["interp_threads"] = a.threads
if a.lds:
    kw["lds_budget_bytes"] = a.lds
mode = nufft.Direct() if a.mode == "direct" else nufft.FastApproximation()
kernel = {"bkb": nufft.BackwardsKaiserBesselKernel, "kb": nufft.KaiserBesselKernel, "gaussian": nufft.GaussianKernel,
          "bspline": nufft.BSplineKernel}[a.kernel]()
plan = nufft.PlanNUFFT(Z, dims, m=a.m, sigma=a.sigma, ntransforms=a.c, kernel_evalmode=mode, kernel=kernel,
                       backend=nufft.ROCBackend(0), synchronise=True, **kw)
info = plan.info()
print(f"plan: Nover={plan.oversampled_dims} bins={[info.bin_dims[d] for d in range(a.dim)]} "
      f"spread_tile={[info.spread_tile[d] for d in range(a.dim)]} x{[info.spread_ntiles[d] for d in range(a.dim)]} "
      f"interp_tile={[info.interp_tile[d] for d in range(a.dim)]} x{[info.interp_ntiles[d] for d in range(a.dim)]} "
      f"lds={info.lds_bytes_spread}/{info.lds_bytes_interp} "
      f"threads={info.spread_threads}/{info.interp_threads} workspace={info.workspace_bytes / 1e9:.2f} GB "
      f"spread_method={info.spread_method} ring_column={list(info.ring_column)} x{info.ring_segments} segments halo={info.ring_halo}", flush=True)

g = torch.Generator(device="cuda").manual_seed(42)
if a.dist == "uniform":
    xs = tuple(torch.rand(Np, dtype=T, device="cuda", generator=g) * (2 * np.pi) for _ in dims)
elif a.dist.startswith("cluster"):      # cluster:<sigma in radians>, centred at pi
    sg = float(a.dist.split(":")[1])
    xs = tuple(torch.randn(Np, dtype=T, device="cuda", generator=g) * sg + np.pi for _ in dims)
else:                                   # "randn": the reference's benchmark protocol (folded N(0, 1) coordinates)
    xs = tuple(torch.randn(Np, dtype=T, device="cuda", generator=g) for _ in dims)
vs = tuple(torch.randn(Np, dtype=Z, device="cuda", generator=g) for _ in range(a.c))
us = tuple(torch.empty(plan.shape, dtype=plan.eltype, device="cuda") for _ in range(a.c))
out = tuple(torch.empty(Np, dtype=Z, device="cuda") for _ in range(a.c))

acc = {}
for rep in range(a.reps + 1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nufft.set_points(plan, xs)
    nufft.exec_type1(us, plan, vs)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    nufft.exec_type2(out, plan, us)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    tm = plan.timer
    tm["wall_sp+t1"] = (t1 - t0) * 1e3
    tm["wall_t2"] = (t2 - t1) * 1e3
    if rep > 0:
        for k, v in tm.items():
            acc.setdefault(k, []).append(v)
print("stage medians (ms):")
for k, v in acc.items():
    print(f"  {k:16s} {np.median(v):9.3f}  (min {np.min(v):.3f})")
t1e = sum(np.median(acc[k]) for k in ("t1_spread", "t1_fft", "t1_deconv"))
t2e = sum(np.median(acc[k]) for k in ("t2_deconv_pad", "t2_fft", "t2_interp"))
sp = np.median(acc["set_points"])
print(f"type-1 exec {t1e:.3f} ms -> {Np / t1e / 1e6:.3f} Gpts/s ; with set_points {Np / (t1e + sp) / 1e6:.3f} Gpts/s")
print(f"type-2 exec {t2e:.3f} ms -> {Np / t2e / 1e6:.3f} Gpts/s ; with set_points {Np / (t2e + sp) / 1e6:.3f} Gpts/s")
print(f"engines: spread {plan.spread_engine_used()}, interp {plan.interp_engine_used()}, column-layer sort {plan.sort_columns_used()} (plan: {list(info.sort_column)})")
