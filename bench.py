#!/usr/bin/env python3
"""Benchmark of the NUFFT hot path on MI355X (driver contract: one JSON line on rank 0).

Workload (BASELINE.json configs[1], "C2"): 3-D, Ns = 256^3, Np = 1e7 uniform-random points, Float64
real data, m = HalfSupport(4), sigma = 2 (plan default) -> oversampled grid 512^3.

A *step* follows the reference's published protocol (benchmark/CPU+AMDGPU/run_benchmarks.jl:80-90):
``set_points!`` + ``exec_type1!`` on inputs already resident in HBM.  ``value`` = whole-job NU-points/s
of K such steps (max over ranks).  The same protocol for type-2 is timed in a second region and
reported under ``type2``.  Stage times come from HIP events recorded on the launch stream *inside*
the timed region (torch events on the current stream, which is the stream handed to the C ABI).

N > 1: one process per GPU (torchrun), every rank owns an independent C2 problem with its own seed
(BASELINE configs[4]: batch of independent plans, one per GPU; weak scaling).  No collective on the
data path; the only RCCL call is the gather of the output spectra to rank 0, issued on a side stream
and overlapped with the next step.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s measured copy rate


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=256, help="uniform grid size per dimension")
    ap.add_argument("--np", type=float, default=1e7, help="non-uniform points per GPU")
    ap.add_argument("--m", type=int, default=4)
    ap.add_argument("--sigma", type=float, default=2.0)
    ap.add_argument("--evalmode", default="fast", choices=["direct", "fast"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--no-reference-protocol", action="store_true")
    ap.add_argument("--force-distributed", action="store_true",
                    help="take the multi-process code path (RCCL init, side-stream gather) even with one rank: "
                         "a single-GPU self-test of the N > 1 path")
    return ap.parse_args()


def algorithmic_bytes(Np, Nover, Nout):
    """SURVEY.md §8(d), Float64 real data, D = 3 (bytes per launch / per transform)."""
    G = float(np.prod(Nover)) * 8                      # oversampled real grid
    S = float((Nover[0] // 2 + 1) * Nover[1] * Nover[2]) * 16
    Oo = float(np.prod(Nout)) * 16
    P = Np * (3 * 8 + 8 + 4)                           # coordinates + value + permutation index
    return {
        # SURVEY §8(d) "type-1 zero + spread" = W(G) zero + R(points) + RMW(G) flush = 3G + P.  The
        # output-driven spreading kernel performs that whole stage in one launch (it writes every grid
        # cell once and needs no zero fill), so its own compulsory traffic is only G + P
        # ("spread_kernel_min"); both figures are reported.
        "spread_kernel": 3 * G + P,
        "spread_kernel_min": G + P,
        "interp_kernel": G + P,                        # R(G) + R(coords) + W(values)
        "type1_exec": 3 * G + P + (G + S) + 2 * Oo,    # zero + spread, FFT (single-pass ideal), deconv (SURVEY 5.96 GB)
        "type2_exec": (S + 2 * Oo) + (S + G) + (G + P),
        "set_points": 2.0 * Np * 3 * 8,
    }


def pmc_traffic(kernel_substr):
    """HBM bytes per launch of a kernel from the newest committed PMC summary (profiles/*_traffic.json,
    written by scripts/summarize_profile.py from separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`
    passes of this same command, with the gfx950 FETCH_SIZE correction).  Counters cannot be collected
    inside the timed run; None if no summary is present."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
    for f in reversed(files):
        try:
            ks = json.load(open(f))["kernels"]
        except Exception:
            continue
        for name, v in ks.items():
            if kernel_substr in name:
                return float(v["hbm_bytes_per_launch"]), os.path.basename(f)
    return None, None


def launch_ranks(a):
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) with torch.distributed.run
    as a CHILD process — before this process has touched the GPU — relay their output (rank 0 prints the
    JSON line) and exit with the child's code.  Never re-executes a process that has initialised HIP."""
    import socket
    import subprocess
    n_visible = torch.cuda.device_count()          # does not initialise the GPU on this image
    if n_visible < a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus}: only {n_visible} GPU(s) visible on this node; refusing to run "
                         f"a smaller job under that name")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def main():
    a = parse()
    if "RANK" not in os.environ and (a.gpus > 1 or a.force_distributed):
        launch_ranks(a)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {a.gpus} bench.py --gpus {a.gpus} ...)")
    distributed = world > 1 or a.force_distributed
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    from nufft_pkg import nufft

    Np = int(a.np)
    dims = (a.n, a.n, a.n)
    mode = nufft.Direct() if a.evalmode == "direct" else nufft.FastApproximation()
    plan = nufft.PlanNUFFT(torch.float64, dims, m=a.m, sigma=a.sigma, kernel_evalmode=mode,
                           backend=nufft.ROCBackend(local_rank))
    info = plan.info()
    g = torch.Generator(device=dev).manual_seed(42 + rank)
    xs = tuple(torch.rand(Np, dtype=torch.float64, device=dev, generator=g) * (2 * np.pi) for _ in dims)
    vp = torch.randn(Np, dtype=torch.float64, device=dev, generator=g)
    uhat = [torch.empty(plan.shape, dtype=torch.complex128, device=dev) for _ in range(2)]   # double buffer
    vout = torch.empty(Np, dtype=torch.float64, device=dev)

    lib, C = nufft.lib, __import__("ctypes")
    from nonuniformffts_jl_amd.plan import _check, _ptr_table

    gather_stream = torch.cuda.Stream(device=dev) if distributed and not a.no_gather else None
    gather_list = None
    if gather_stream is not None and rank == 0:
        # complex spectra travel as their (re, im) real views (same bytes; every backend supports reals)
        gather_list = [[torch.empty_like(torch.view_as_real(uhat[0])) for _ in range(world)] for _ in range(2)]
    gather_done = [None, None]

    def stream_ptr():
        return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def step_type1(k, events=None):
        """set_points! + exec_type1! (the four stages of src/NonuniformFFTs.jl:157-186, called one by one
        so that HIP events can be recorded between them on the launch stream)."""
        out = uhat[k % 2]
        if gather_done[k % 2] is not None:    # the gather that still reads this buffer must be done
            torch.cuda.current_stream(dev).wait_event(gather_done[k % 2])
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)] if events is not None else None
        s = stream_ptr()
        if ev: ev[0].record()
        nufft.set_points(plan, xs)
        if ev: ev[1].record()
        if ev: ev[2].record()      # "(0) fill with zeros" no longer exists: the spreading kernel writes every cell
        _check(lib.nufft_spread(plan._handle, _ptr_table((vp,)), s))
        if ev: ev[3].record()
        _check(lib.nufft_fft_forward(plan._handle, s))
        if ev: ev[4].record()
        _check(lib.nufft_deconvolve_truncate(plan._handle, _ptr_table((out,)), s))
        if ev: ev[5].record()
        if events is not None:
            events.append(ev)
        if gather_stream is not None:
            done = torch.cuda.Event()
            done.record()
            gather_stream.wait_event(done)
            with torch.cuda.stream(gather_stream):
                import torch.distributed as dist
                dist.gather(torch.view_as_real(out), gather_list[k % 2] if rank == 0 else None, dst=0)   # stream-ordered, host does not block
                e = torch.cuda.Event()
                e.record()
                gather_done[k % 2] = e

    def step_type2(k, events=None):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)] if events is not None else None
        s = stream_ptr()
        if ev: ev[0].record()
        nufft.set_points(plan, xs)
        if ev: ev[1].record()
        _check(lib.nufft_deconvolve_pad(plan._handle, _ptr_table((uhat[0],)), s))
        if ev: ev[2].record()
        _check(lib.nufft_fft_backward(plan._handle, s))
        if ev: ev[3].record()
        _check(lib.nufft_interpolate(plan._handle, _ptr_table((vout,)), s))
        if ev: ev[4].record()
        if events is not None:
            events.append(ev)

    def barrier():
        torch.cuda.synchronize(dev)
        if distributed:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed(step_fn, K, W):
        for k in range(W):
            step_fn(k)
        events = []
        barrier()
        t0 = time.perf_counter()
        for k in range(K):
            step_fn(k, events)
        if gather_stream is not None:
            gather_stream.synchronize()
        barrier()
        dt = time.perf_counter() - t0
        if distributed:
            import torch.distributed as dist
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, events

    def stage_ms(events, names):
        out = {}
        for i, name in enumerate(names):
            vals = [ev[i].elapsed_time(ev[i + 1]) for ev in events]
            out[name] = float(np.mean(vals))
        return out

    dt1, ev1 = timed(step_type1, a.steps, a.warmup)
    st1 = stage_ms(ev1, ["set_points", "zero", "spread", "fft", "deconv"])
    st1.pop("zero")      # empty interval: the spreading kernel writes every cell, no fill_with_zeros stage
    gs_save, gather_stream = gather_stream, None      # type-2 region has no gather
    dt2, ev2 = timed(step_type2, a.steps, a.warmup)
    st2 = stage_ms(ev2, ["set_points", "deconv_pad", "fft", "interp"])
    gather_stream = gs_save

    ab = algorithmic_bytes(Np, plan.oversampled_dims, plan.size)
    value = world * Np * a.steps / dt1
    value2 = world * Np * a.steps / dt2
    spread_s = st1["spread"] * 1e-3
    exec1_ms = st1["spread"] + st1["fft"] + st1["deconv"]
    exec2_ms = st2["deconv_pad"] + st2["fft"] + st2["interp"]
    traffic_b, traffic_src = pmc_traffic("spread_tile_kernel<double, false, 3, 4")
    traffic_gb = traffic_b / 1e9 if traffic_b is not None else None
    result = {
        "metric": "NU-points/s, type-1 NUFFT (set_points! + exec_type1!), 256^3 Float64 m=4",
        "value": value,
        "unit": "NU-points/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": dt1 / a.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"C2: 3-D type-1+type-2, Ns={a.n}^3, Np={Np:.0e} uniform-random per GPU, Float64 real, "
                        f"m={a.m}, sigma={a.sigma} (oversampled {plan.oversampled_dims}), "
                        f"{'Direct' if a.evalmode == 'direct' else 'FastApproximation'} window",
            "protocol": "set_points! + exec_type1! per step, inputs resident in HBM (reference benchmark protocol)",
            "spread_tile": [int(info.spread_tile[d]) for d in range(3)],
            "interp_tile": [int(info.interp_tile[d]) for d in range(3)],
            "lds_bytes": int(info.lds_bytes_spread),
            "parallelism": f"{world} independent plan(s), one per GPU" + ("" if world == 1 or a.no_gather else "; RCCL gather of spectra to rank 0 overlapped on a side stream"),
        },
        "roofline": {
            "bound": "hbm",
            "kernel": "spread_tile_kernel<double,false,3,4,false> (zero + spread stage, one launch)",
            "achieved": ab["spread_kernel"] / spread_s / 1e9,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": ab["spread_kernel"] / spread_s / 1e9 / HBM_PEAK_GBS,
            "traffic": traffic_gb,
            "traffic_unit": "GB per launch (PMC FETCH_SIZE x2 + WRITE_SIZE)",
            "traffic_source": traffic_src,
            "binding_resource": "LDS float atomics (ds_add_f64, 8.5 cycles per wave instruction per CU) and the "
                                "scalar/vector issue of the clipped stencil loop, not HBM: see DESIGN.md section 4.2",
            "algorithmic_bytes_per_launch": ab["spread_kernel"],
            "min_traffic_bytes_per_launch": ab["spread_kernel_min"],
            "achieved_min_traffic": ab["spread_kernel_min"] / spread_s / 1e9,
            "kernel_ms": st1["spread"],
            "type1_exec_frac": ab["type1_exec"] / (exec1_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "type2_exec_frac": ab["type2_exec"] / (exec2_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
        },
        "type1": {"stages_ms": st1, "exec_only_pts_per_s": Np / (exec1_ms * 1e-3), "with_set_points_pts_per_s": value},
        "type2": {"stages_ms": st2, "exec_only_pts_per_s": Np / (exec2_ms * 1e-3), "with_set_points_pts_per_s": value2,
                  "ms_per_step": dt2 / a.steps * 1e3},
    }

    if world == 1 and not a.no_reference_protocol:
        result["reference_protocol"] = reference_protocol(a, nufft, dev)
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(a)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if distributed:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def reference_protocol(a, nufft, dev):
    """The reference's published benchmark protocol, for comparison with BASELINE.md (not the headline metric):
    sigma = 1.5, BackwardsKaiserBessel with Direct() evaluation (the ROC defaults), coordinates ~ N(0, 1) folded
    into the period, time = set_points! + exec! (benchmark/CPU+AMDGPU/run_benchmarks.jl:57-90), same Ns and Np."""
    Np = int(a.np)
    dims = (a.n, a.n, a.n)
    plan = nufft.PlanNUFFT(torch.float64, dims, m=a.m, sigma=1.5, kernel_evalmode=nufft.Direct(), backend=nufft.ROCBackend(dev.index or 0))
    g = torch.Generator(device=dev).manual_seed(4242)
    xs = tuple(torch.randn(Np, dtype=torch.float64, device=dev, generator=g) for _ in dims)
    v = torch.randn(Np, dtype=torch.float64, device=dev, generator=g)
    u = torch.empty(plan.shape, dtype=torch.complex128, device=dev)
    out = torch.empty(Np, dtype=torch.float64, device=dev)
    res = {}
    for name, fn in (("type1", lambda: nufft.exec_type1(u, plan, v)), ("type2", lambda: nufft.exec_type2(out, plan, u))):
        for _ in range(2):
            nufft.set_points(plan, xs); fn()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        K = 10
        for _ in range(K):
            nufft.set_points(plan, xs); fn()
        torch.cuda.synchronize(dev)
        res[name + "_pts_per_s"] = Np * K / (time.perf_counter() - t0)
    res["config"] = f"Ns={a.n}^3, Np={Np:.0e} ~ N(0,1) folded, Float64, m={a.m}, sigma=1.5 (oversampled {plan.oversampled_dims}), Direct window, set_points! + exec!"
    res["published_mi300a_pts_per_s"] = {"type1": "2.4e8-2.7e8", "type2": "6.2e8-9.6e8", "source": "BASELINE.md (Np = 1.7e7 ... 1.7e8)"}
    return res


def cpu_baseline(a):
    """The oracle's C restatement of the reference's blocked CPU algorithm (+ pocketfft), timed on the
    host cores of this box on a bounded sample of the same workload (kind = "port": the reference's
    Julia CPU backend cannot run here — no Julia runtime)."""
    try:
        from oracle import c_oracle as CO, nufft_oracle as O
        if not CO.available():
            return {"value": None, "unit": "NU-points/s", "cores": 0, "kind": "port", "sample": "oracle/libnufft_oracle.so not built"}
        cores = CO.num_threads()
        dims = (a.n, a.n, a.n)
        oplan = O.OraclePlan(dims, is_real=True, M=a.m, sigma=a.sigma, evalmode=O.FAST_APPROXIMATION)
        rng = np.random.default_rng(42)

        def run(Np):
            xs = [rng.random(Np) * O.TWO_PI for _ in dims]
            v = rng.standard_normal(Np)
            t0 = time.perf_counter()
            O.set_points(oplan, xs)
            CO.exec_type1(oplan, v)
            return time.perf_counter() - t0

        t_probe = run(200_000)                  # dominated by the 512^3 FFT: the fixed cost
        t_mid = run(1_000_000)
        per_pt = max((t_mid - t_probe) / 800_000, 1e-9)
        Np_s = int(min(a.np, max(1_000_000, (12.0 - t_probe) / per_pt)))
        t = run(Np_s)
        return {"value": Np_s / t, "unit": "NU-points/s", "cores": cores, "kind": "port",
                "sample": f"one set_points+type-1 transform, same grid ({a.n}^3, sigma={a.sigma}, m={a.m}), "
                          f"Np={Np_s} of {int(a.np)} points, {t:.1f} s wall; C/OpenMP blocked spreading (plain adds under a lock, "
                          f"the reference's default) + scipy pocketfft; threads = CPUs available to the process "
                          f"(affinity capped by the cgroup quota: {CO.available_cpus()} of {os.cpu_count()} logical CPUs)",
                "seconds": t}
    except Exception as exc:  # the baseline is informative only
        return {"value": None, "unit": "NU-points/s", "cores": 0, "kind": "port", "sample": f"failed: {exc!r}"}


if __name__ == "__main__":
    main()
