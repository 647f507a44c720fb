#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ from the CPU oracle (oracle/nufft_oracle.py).

The reference ships no golden vectors and cannot run here (Julia, no runtime), so these fixtures are
produced by the oracle — which is itself pinned against the reference's known-answer tests
(tests/test_oracle_kats.py) — and every fixture also stores the exact NUDFT of its inputs, so a reader
can check the vectors against the analytic definition without trusting the oracle.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import nufft_oracle as O  # noqa: E402


def make(name, dims, Z, M, sigma, evalmode, Np, seed, ntransforms=1, special_points=None, kernel=O.KERNEL_BKB):
    Z = np.dtype(Z)
    is_real = Z.kind == "f"
    T = np.float32 if Z in (np.dtype(np.float32), np.dtype(np.complex64)) else np.float64
    rng = np.random.default_rng(seed)
    xs = [((rng.random(Np) * 3 - 1) * O.TWO_PI).astype(T) for _ in dims]
    if special_points is not None:
        for d in range(len(dims)):
            xs[d][: len(special_points)] = np.asarray(special_points, dtype=T)
    if is_real:
        vs = [rng.standard_normal(Np).astype(Z) for _ in range(ntransforms)]
    else:
        vs = [(rng.standard_normal(Np) + 1j * rng.standard_normal(Np)).astype(Z) for _ in range(ntransforms)]
    plan = O.OraclePlan(dims, is_real=is_real, dtype=T, M=M, sigma=sigma, evalmode=evalmode, ntransforms=ntransforms,
                        kernel=kernel)
    O.set_points(plan, xs)
    shape = tuple(reversed(plan.size))
    ws = [(rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(plan.cdtype) for _ in range(ntransforms)]
    t1 = O.exec_type1(plan, vs if ntransforms > 1 else vs[0])
    t2 = O.exec_type2(plan, ws if ntransforms > 1 else ws[0])
    t1 = t1 if ntransforms > 1 else [t1]
    t2 = t2 if ntransforms > 1 else [t2]
    x64 = [x.astype(np.float64) for x in xs]
    exact1 = [O.nudft_type1(plan.ks, x64, v) for v in vs]
    exact2 = [O.nudft_type2_real(plan, x64, w) if is_real else O.nudft_type2(plan.ks, x64, w) for w in ws]
    out = dict(dims=np.array(dims), M=M, sigma=sigma, evalmode=evalmode, is_real=is_real, ntransforms=ntransforms,
               dtype=str(Z), nover=np.array(plan.Nover), kernel=kernel)
    for d, x in enumerate(xs):
        out[f"x{d}"] = x
    for c in range(ntransforms):
        out[f"v{c}"] = vs[c]
        out[f"w{c}"] = ws[c]
        out[f"type1_{c}"] = t1[c]
        out[f"type2_{c}"] = t2[c]
        out[f"exact1_{c}"] = exact1[c].astype(np.complex128)
        out[f"exact2_{c}"] = np.asarray(exact2[c])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    e1 = max(O.l2_error(t1[c], exact1[c]) for c in range(ntransforms))
    e2 = max(O.l2_error(t2[c], exact2[c]) for c in range(ntransforms))
    print(f"{name}: Nover={plan.Nover} type-1 err vs exact {e1:.2e}, type-2 {e2:.2e}")


if __name__ == "__main__":
    edge = [np.nextafter(O.TWO_PI, 0.0), np.nextafter(np.pi, 0.0), 0.0, -0.0, O.TWO_PI, -O.TWO_PI, 3 * O.TWO_PI + 0.1]
    # BASELINE config C1: 1-D type-1, N = 256, Np = 1e4, Float64, m = 4 (sigma = 2 plan default)
    make("c1_1d_f64_m4", (256,), np.float64, 4, 2.0, O.FAST_APPROXIMATION, 10_000, seed=0)
    make("c1_1d_f64_m4_direct", (256,), np.float64, 4, 2.0, O.DIRECT, 10_000, seed=0)
    # tiny 3-D cases of the four element types, with edge points mixed in
    make("tiny3d_f64", (12, 16, 10), np.float64, 4, 2.0, O.DIRECT, 500, seed=1, special_points=edge)
    make("tiny3d_c128", (12, 16, 10), np.complex128, 4, 1.5, O.FAST_APPROXIMATION, 500, seed=2, special_points=edge)
    make("tiny3d_f32", (12, 16, 10), np.float32, 4, 2.0, O.DIRECT, 500, seed=3)
    make("tiny3d_c64", (12, 16, 10), np.complex64, 4, 2.0, O.FAST_APPROXIMATION, 500, seed=4)
    make("tiny3d_f64_nt3", (12, 16, 10), np.float64, 4, 2.0, O.DIRECT, 400, seed=5, ntransforms=3)
    make("small2d_c128_m6", (20, 27), np.complex128, 6, 2.0, O.DIRECT, 600, seed=6, special_points=edge)
    make("small1d_c128_m8", (32,), np.complex128, 8, 1.5, O.FAST_APPROXIMATION, 64, seed=7, special_points=edge)
    # the other spreading kernels
    make("kb_3d_f64_m4", (12, 16, 10), np.float64, 4, 2.0, O.DIRECT, 400, seed=8, kernel=O.KERNEL_KB, special_points=edge)
    make("kb_2d_c64_m4_fast", (20, 27), np.complex64, 4, 2.0, O.FAST_APPROXIMATION, 400, seed=9, kernel=O.KERNEL_KB)
    make("gaussian_3d_f64_m6", (12, 16, 10), np.float64, 6, 2.0, O.FAST_APPROXIMATION, 400, seed=10, kernel=O.KERNEL_GAUSSIAN,
         special_points=edge)
    make("gaussian_1d_c128_m8_direct", (64,), np.complex128, 8, 2.0, O.DIRECT, 200, seed=11, kernel=O.KERNEL_GAUSSIAN)
    make("bspline_3d_f64_m6", (12, 16, 10), np.float64, 6, 2.0, O.DIRECT, 400, seed=12, kernel=O.KERNEL_BSPLINE,
         special_points=edge)
    make("bspline_2d_f32_m4", (20, 27), np.float32, 4, 2.0, O.FAST_APPROXIMATION, 400, seed=13, kernel=O.KERNEL_BSPLINE)
