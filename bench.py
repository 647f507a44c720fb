#!/usr/bin/env python3
"""Benchmark of the NUFFT hot path on MI355X (driver contract: one JSON line on rank 0).

Workloads (BASELINE.json ``configs``; ``--config``):
  c2 (default, the configuration the metric is quoted on): 3-D type-1 + type-2, Ns = 256^3, Np = 1e7 uniform-random
      points, Float64 real data, m = HalfSupport(4), sigma = 2 (plan default) -> oversampled grid 512^3
  c3: 3-D type-1, Ns = 512^3, Np = 1e8, ComplexF32, m = 8 (oversampled 1024^3)
  c4: c2 with ntransforms = 3 (three value vectors spread / interpolated with one set of points)
  (configs[0] is the reference's CPU-only plumbing case and configs[4] = c2 on N GPUs: ``--gpus N``.)

A *step* follows the reference's published protocol (benchmark/CPU+AMDGPU/run_benchmarks.jl:80-90):
``set_points!`` + ``exec_type1!`` on inputs already resident in HBM.  ``value`` = whole-job NU-points/s of K such
steps (max over ranks).  BASELINE.json does not fix the window evaluation, and the reference has two defaults:
``FastApproximation()`` on its CPU backend and ``Direct()`` on ROC (ext/NonuniformFFTsAMDGPUExt.jl:56).  Both are measured
and reported as equally complete records (value, stage times, type 2, roofline fraction): the top-level ``value`` is the
polynomial window (continuity with round 1's line), the sibling record ``direct`` the ROC default.  Type-2 is timed the
same way in a second region.  Stage times come from HIP events recorded on the launch stream *inside* the timed region (torch events
on the current stream, which is the stream handed to the C ABI).

N > 1 (``--gpus N``; without a launcher this script starts the N ranks itself): one process per GPU, every rank
owns an independent problem with its own seed (BASELINE configs[4]: batch of independent plans, one per GPU; weak
scaling).  No collective on the data path; the only RCCL call is the gather of the output spectra to rank 0,
issued on a side stream and overlapped with the next step.
"""
import argparse
import json
import os
import re
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP32_PEAK_TFLOPS = 157.3   # MI355X FP32 vector / matrix peak (256 CUs x 4 SIMDs x 32 lanes x 2 flop x 2.4 GHz)

CONFIGS = {
    "c2": dict(Z="float64", n=256, np=1e7, m=4, sigma=2.0, C=1, steps=20,
               label="C2: 3-D type-1+type-2, Ns=256^3, Np=1e7 uniform-random per GPU, Float64 real"),
    "c3": dict(Z="complex64", n=512, np=1e8, m=8, sigma=2.0, C=1, steps=5,
               label="C3: 3-D type-1, Ns=512^3, Np=1e8 uniform-random, ComplexF32 (high accuracy, LDS pressure)"),
    "c4": dict(Z="float64", n=256, np=1e7, m=4, sigma=2.0, C=3, steps=10,
               label="C4: 3-D ntransforms=3, Ns=256^3, Np=1e7 uniform-random, Float64 real (vector-valued simultaneous spread)"),
}


# NUFFT_BENCH_SHARE_GPU=1: run the N ranks of `--gpus N` on ONE device with the gloo backend (no gather).  A builder's self-test of
# the launcher, the barriers and the MAX-over-ranks timing on a one-GPU box; the JSON line carries "shared_gpu_self_test": true.
SHARE_GPU_SELF_TEST = os.environ.get("NUFFT_BENCH_SHARE_GPU", "0") == "1"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps (0: the configuration's default)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--n", type=int, default=0, help="override: uniform grid size per dimension")
    ap.add_argument("--np", type=float, default=0, help="override: non-uniform points per GPU")
    ap.add_argument("--m", type=int, default=0)
    ap.add_argument("--sigma", type=float, default=0)
    ap.add_argument("--evalmode", default="direct", choices=["direct", "fast"],
                    help="window evaluation of the top-level value (default: Direct(), the reference's default on ROCBackend, "
                         "ext/NonuniformFFTsAMDGPUExt.jl:56 — since round 5; rounds 1-4 led with FastApproximation; the other mode is "
                         "measured too and reported as an equally complete sibling record, and as config.fast_value / config.direct_value)")
    ap.add_argument("--only-headline", action="store_true", help="skip the sibling evaluation mode, the reference "
                    "protocol, the density sweep, the HBM probe and the CPU baseline (profiling runs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--no-reference-protocol", action="store_true")
    ap.add_argument("--no-density-sweep", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the compact C3 / C4 records of the default run")
    ap.add_argument("--shard-components", action="store_true",
                    help="configurations with ntransforms = C > 1 (c4): the C components of ONE transform are sharded over the ranks "
                         "(component c on rank c mod N, the same points on every rank; strong scaling) instead of one independent "
                         "C-component problem per rank")
    ap.add_argument("--force-distributed", action="store_true",
                    help="take the multi-process code path (RCCL init, side-stream gather) even with one rank: "
                         "a single-GPU self-test of the N > 1 path")
    return ap.parse_args()


def algorithmic_bytes(Np, Nover, Nout, is_complex, real_bytes, C):
    """SURVEY.md §8(d) (bytes per launch / per transform), D = 3: G = oversampled grid of one component, S = its
    spectrum, O = output, P = point data (coordinates + value + permutation index)."""
    zb = real_bytes * (2 if is_complex else 1)
    G = float(np.prod(Nover)) * zb
    S = float(np.prod(Nover)) * 2 * real_bytes if is_complex else float((Nover[0] // 2 + 1) * Nover[1] * Nover[2]) * 2 * real_bytes
    Oo = float(np.prod(Nout)) * 2 * real_bytes
    Pc = Np * (3 * real_bytes + 4)                     # coordinates + permutation index: once
    Pv = Np * zb                                       # value: per component
    return {
        # SURVEY §8(d) "type-1 zero + spread" = W(G) zero + R(points) + RMW(G) flush = 3G + P per component.  Both
        # spreading engines perform that whole stage in one launch per component, write every grid cell once and need
        # no zero fill, so their own compulsory traffic is G + P ("spread_kernel_min"); both figures are reported.
        "spread_kernel": C * (3 * G + Pv) + Pc,
        "spread_kernel_min": C * (G + Pv) + Pc,
        "interp_kernel": C * (G + Pv) + Pc,            # R(G) + R(coords) + W(values)
        "type1_exec": C * (3 * G + Pv + (G + S) + 2 * Oo) + Pc,    # zero + spread, FFT (single-pass ideal), deconv
        "type2_exec": C * ((S + 2 * Oo) + (S + G) + (G + Pv)) + Pc,
        "set_points": 2.0 * Np * 3 * real_bytes,
        "G": G,
    }


def profile_files(config, suffix, evalmode=None):
    """Committed profile summaries of one configuration, oldest first: profiles/round<N>_<letter>_bench_<config>[_direct]<suffix>.
    The profiles are collected per window mode (`..._<config>_direct_*` = Direct(), `..._<config>_*` = the polynomial window);
    with `evalmode` ("Direct" / "FastApproximation") only that mode's files are returned, so a kernel whose name does not
    carry the mode (the interpolation ring) is never looked up in the other mode's profile."""
    import glob
    out = []
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*" + suffix))):
        b = os.path.basename(f)
        if f"_{config}_" not in b and f"_{config}." not in b:
            continue
        is_direct = f"_{config}_direct" in b
        if evalmode is not None and is_direct != (evalmode == "Direct"):
            continue
        out.append(f)
    # newest round / letter last: round10 sorts behind round9
    def key(f):
        mm = re.match(r"round(\d+)_([a-z]+)_", os.path.basename(f))
        return (int(mm.group(1)), mm.group(2)) if mm else (0, "")
    return sorted(out, key=key)


def pmc_traffic(kernel_substr, config, evalmode=None):
    """HBM bytes per launch of a kernel from the newest committed PMC summary of this configuration AND window mode
    (profiles/*_traffic.json, written by scripts/summarize_profile.py from separate `rocprofv3 --pmc FETCH_SIZE` /
    `--pmc WRITE_SIZE` passes of this same command, with the gfx950 FETCH_SIZE correction).  Counters cannot be collected
    inside the timed run; None if no summary is present."""
    for f in reversed(profile_files(config, "_traffic.json", evalmode)):
        try:
            ks = json.load(open(f))["kernels"]
        except Exception:
            continue
        for name, v in ks.items():
            if kernel_substr in name:
                return float(v["hbm_bytes_per_launch"]), os.path.basename(f), v.get("kernel_avg_us")
    return None, None, None


def sq_counters(kernel_substr, config, evalmode=None):
    """SQ counters per launch of a kernel from the newest committed profiles/*_sq.json of this configuration and window mode (written by
    scripts/summarize_profile.py from the SQ passes of scripts/profile_bench.sh), with the derived figures that name the binding
    resource; None if no such profile is committed.  Nothing here is typed in by hand."""
    for f in reversed(profile_files(config, "_sq.json", evalmode)):
        try:
            ks = json.load(open(f))["kernels"]
        except Exception:
            continue
        for name, v in ks.items():
            if kernel_substr in name:
                d = {"source": os.path.basename(f), "kernel": name, "counters": v}
                wc = v.get("SQ_WAVE_CYCLES")
                if wc:
                    # (ratios between SQ counters of one pass are what is meaningful: instruction mix, active / wait shares of the wave cycles)
                    for key, label in (("SQ_ACTIVE_INST_VALU", "valu_active_share_of_wave_cycles"), ("SQ_ACTIVE_INST_LDS", "lds_active_share_of_wave_cycles"),
                                       ("SQ_WAIT_INST_LDS", "wait_lds_share_of_wave_cycles")):
                        if key in v:
                            d[label] = v[key] / wc
                if v.get("SQ_LDS_IDX_ACTIVE"):
                    d["lds_bank_conflict_share_of_lds_active"] = v.get("SQ_LDS_BANK_CONFLICT", 0.0) / v["SQ_LDS_IDX_ACTIVE"]
                if v.get("SQ_WAVES") and v.get("SQ_INSTS_LDS") is not None:
                    d["lds_instructions_per_wave"] = v["SQ_INSTS_LDS"] / v["SQ_WAVES"]
                if v.get("SQ_WAVES") and v.get("SQ_INSTS_VALU") is not None:
                    d["valu_instructions_per_wave"] = v["SQ_INSTS_VALU"] / v["SQ_WAVES"]
                return d
    return None


# The driver's parser keeps the first 20 SCALAR keys of `config` (lists and records are skipped).  These lead, in this order, so that the
# other single-GPU BASELINE configurations (C3, C4), the two window modes and the reference protocol are part of the driver-observed line
# (round 5 lost c3_* / c4_* behind stage times and sort keys; tests/test_bench_cli.py pins this list).
CONFIG_LEAD_KEYS = ("workload", "direct_value", "fast_value", "type2_value",
                    "c3_value", "c3_type2_value", "c3_spread_ms", "c3_interp_ms", "c3_set_points_ms", "c3_fp32_frac", "c3_direct_value",
                    "c4_value", "c4_type2_value", "c4_spread_ms", "c4_interp_ms", "c4_direct_value",
                    "refproto_f64_type1_value", "refproto_f64_type2_value", "refproto_c128_type1_value", "workspace_bytes")


def lead_config(config):
    """`config` with CONFIG_LEAD_KEYS first (those that are present), everything else behind in its original order."""
    out = {k: config[k] for k in CONFIG_LEAD_KEYS if k in config}
    out.update((k, v) for k, v in config.items() if k not in out)
    return out


def first_scalar_keys(config, n=20):
    """What the driver's parser keeps of `config`: its first n scalar-valued keys."""
    return [k for k, v in config.items() if not isinstance(v, (dict, list, tuple))][:n]


def launch_ranks(a):
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) with torch.distributed.run
    as a CHILD process — before this process has touched the GPU — relay their output (rank 0 prints the
    JSON line) and exit with the child's code.  Never re-executes a process that has initialised HIP."""
    import socket
    import subprocess
    n_visible = torch.cuda.device_count()          # does not initialise the GPU on this image
    if n_visible < a.gpus and not SHARE_GPU_SELF_TEST:
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


def sync_barrier(dev, distributed):
    """Device sync + (N > 1) barrier + device sync: both ends of the timed region (driver contract)."""
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    if distributed:
        import torch.distributed as dist
        dist.barrier()
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)


def timed_steps(step_fn, K, W, dev, distributed, after=None):
    """W untimed warm-up steps, then exactly K timed steps bracketed by sync_barrier; returns (seconds = MAX over ranks,
    events).  `step_fn(k, events)` runs one step (events is None during warm-up); `after()` runs inside the timed region
    behind the last step (e.g. waiting for the side-stream gather)."""
    for k in range(W):
        step_fn(k, None)
    events = []
    sync_barrier(dev, distributed)
    t0 = time.perf_counter()
    for k in range(K):
        step_fn(k, events)
    if after is not None:
        after()
    sync_barrier(dev, distributed)
    dt = time.perf_counter() - t0
    if distributed:
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, events


def gather_components(outs, recv, rank, dst=0):
    """The single collective of the path: every component's output spectrum of every rank to rank `dst`.
    outs: C tensors of this rank; recv: on rank dst, C lists of world-size receive buffers (None elsewhere).  Complex
    spectra travel as their (re, im) real views (same bytes; every backend supports reals)."""
    import torch.distributed as dist
    for c, o in enumerate(outs):
        src = torch.view_as_real(o) if o.is_complex() else o
        dist.gather(src, recv[c] if rank == dst else None, dst=dst)


def hbm_probe(dev):
    """Measured HBM roofline of this box (device-to-device copy / triad over buffers far larger than the
    256 MB infinity cache).  GB/s of algorithmic bytes: copy moves 2 x, triad 3 x the buffer.  (The stand-alone
    hand-written version of the same probe is scripts/hbm_roofline.hip; its output is in profiles/.)"""
    n = (1 << 31) // 8                                  # 2 GiB per buffer
    a_ = torch.empty(n, dtype=torch.float64, device=dev)
    b_ = torch.ones(n, dtype=torch.float64, device=dev)
    c_ = torch.ones(n, dtype=torch.float64, device=dev)
    res = {}
    for name, fn, moved in (("copy", lambda: a_.copy_(b_), 2.0), ("triad", lambda: torch.add(b_, c_, alpha=0.5, out=a_), 3.0)):
        best = 1e30
        for r in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); e1.synchronize()
            if r:
                best = min(best, e0.elapsed_time(e1))
        res[name + "_GBs"] = moved * n * 8 / (best * 1e-3) / 1e9
    del a_, b_, c_
    torch.cuda.empty_cache()
    res["peak_measured_GBs"] = max(res["copy_GBs"], res["triad_GBs"])
    return res


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
    if SHARE_GPU_SELF_TEST:
        local_rank = 0                  # every rank on device 0: a self-test of the N > 1 process logic, never a scaling number
        a.no_gather = True              # gloo has no device gather; RCCL refuses two ranks on one device
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if SHARE_GPU_SELF_TEST:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from nufft_pkg import nufft
    import ctypes as C
    from nonuniformffts_jl_amd.plan import _check, _ptr_table
    lib = nufft.lib

    cfg = dict(CONFIGS[a.config])
    for k in ("n", "np", "m", "sigma"):
        if getattr(a, k):
            cfg[k] = getattr(a, k)
    steps = a.steps if a.steps > 0 else cfg["steps"]
    full = world == 1 and not a.only_headline

    gather_stream = torch.cuda.Stream(device=dev) if distributed and not a.no_gather else None

    def stream_ptr():
        return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def prepare(cfg):
        """Synthetic inputs of one configuration, resident in HBM (U[0, 2pi) coordinates, N(0, 1) values,
        seed 42 + rank)."""
        Z = getattr(torch, cfg["Z"])
        T = torch.float32 if Z in (torch.float32, torch.complex64) else torch.float64
        P = dict(Z=Z, T=T, CT=torch.complex64 if T == torch.float32 else torch.complex128, is_complex=Z.is_complex,
                 real_bytes=4 if T == torch.float32 else 8, Np=int(cfg["np"]), Cn=int(cfg["C"]), dims=(cfg["n"],) * 3, cfg=cfg)
        shard = bool(a.shard_components) and P["Cn"] > 1
        g = torch.Generator(device=dev).manual_seed(42 + (0 if shard else rank))      # (sharded components: one point set for all ranks)
        P["xs"] = tuple(torch.rand(P["Np"], dtype=T, device=dev, generator=g) * (2 * np.pi) for _ in P["dims"])
        P["vps"] = tuple(torch.randn(P["Np"], dtype=Z, device=dev, generator=g) for _ in range(P["Cn"]))
        P["C_total"], P["sharded"] = P["Cn"], shard
        if shard:
            # component c of the transform lives on rank c mod world (nonuniformffts.jl_amd/batch.py: PlanBatch.from_ntransforms)
            owned = list(range(rank, P["Cn"], world))
            if not owned:
                raise SystemExit(f"bench.py --shard-components: {P['Cn']} components cannot occupy {world} ranks")
            P["vps"] = tuple(P["vps"][c] for c in owned)
            P["Cn"] = len(owned)
            P["rounds"] = (P["C_total"] + world - 1) // world       # gathers per step: ranks with fewer components send a placeholder
        return P

    def measure(P, evalmode_name, steps, gather):
        """K steps of set_points! + exec_type1!, then of set_points! + exec_type2!, with stage events."""
        cfg, Z, CT, Np, Cn, dims, xs, vps = P["cfg"], P["Z"], P["CT"], P["Np"], P["Cn"], P["dims"], P["xs"], P["vps"]
        mode = nufft.Direct() if evalmode_name == "direct" else nufft.FastApproximation()
        plan = nufft.PlanNUFFT(Z, dims, m=cfg["m"], sigma=cfg["sigma"], ntransforms=Cn, kernel_evalmode=mode,
                               backend=nufft.ROCBackend(local_rank))
        info = plan.info()
        uhat = [tuple(torch.empty(plan.shape, dtype=CT, device=dev) for _ in range(Cn)) for _ in range(2)]   # double buffer
        vout = tuple(torch.empty(Np, dtype=Z, device=dev) for _ in range(Cn))
        gather_list = None
        n_gather = P.get("rounds", Cn)                     # collectives per step (sharded components: one per round)
        pad = tuple(torch.zeros_like(uhat[0][0]) for _ in range(n_gather - Cn))      # placeholders of ranks that own fewer components
        if gather and gather_stream is not None and rank == 0:
            # complex spectra travel as their (re, im) real views (same bytes; every backend supports reals);
            # one receive buffer per (double buffer, component, source rank)
            gather_list = [[[torch.empty_like(torch.view_as_real(uhat[0][0])) for _ in range(world)] for _ in range(n_gather)] for _ in range(2)]
        gather_done = [None, None]
        use_gather = [gather and gather_stream is not None]

        def step_type1(k, events=None):
            """set_points! + exec_type1! (the stages of src/NonuniformFFTs.jl:157-186, called one by one so that
            HIP events can be recorded between them on the launch stream)."""
            out = uhat[k % 2]
            if gather_done[k % 2] is not None:    # the gather that still reads this buffer must be done
                torch.cuda.current_stream(dev).wait_event(gather_done[k % 2])
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)] if events is not None else None
            s = stream_ptr()
            if ev: ev[0].record()
            nufft.set_points(plan, xs)
            if ev: ev[1].record()
            # writes every grid cell: no fill_with_zeros stage.  The stage as exec_type1 enqueues it: on the ring's halo variant the FFT
            # stage that follows completes the grid (its first pass adds the side buffer of the stencil reach), and is timed as such
            _check(lib.nufft_spread_deferred(plan._handle, _ptr_table(vps), s))
            if ev: ev[2].record()
            _check(lib.nufft_fft_forward(plan._handle, s))
            if ev: ev[3].record()
            _check(lib.nufft_deconvolve_truncate(plan._handle, _ptr_table(out), s))
            if ev: ev[4].record()
            if events is not None:
                events.append(ev)
            if use_gather[0]:
                import torch.distributed as dist
                done = torch.cuda.Event()
                done.record()
                gather_stream.wait_event(done)
                with torch.cuda.stream(gather_stream):
                    # every component's spectrum (stream-ordered, the host does not block)
                    gather_components(tuple(out) + pad, gather_list[k % 2] if rank == 0 else None, rank)
                    e = torch.cuda.Event()
                    e.record()
                    gather_done[k % 2] = e

        def step_type2(k, events=None):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)] if events is not None else None
            s = stream_ptr()
            if ev: ev[0].record()
            nufft.set_points(plan, xs)
            if ev: ev[1].record()
            _check(lib.nufft_deconvolve_pad(plan._handle, _ptr_table(uhat[0]), s))
            if ev: ev[2].record()
            _check(lib.nufft_fft_backward(plan._handle, s))
            if ev: ev[3].record()
            _check(lib.nufft_interpolate(plan._handle, _ptr_table(vout), s))
            if ev: ev[4].record()
            if events is not None:
                events.append(ev)

        def timed(step_fn, K, W):
            def step(k, events):
                step_fn(k, events)
            return timed_steps(step, K, W, dev, distributed,
                               after=(gather_stream.synchronize if use_gather[0] else None))

        def stage_ms(events, names):
            return {name: float(np.mean([ev[i].elapsed_time(ev[i + 1]) for ev in events])) for i, name in enumerate(names)}

        dt1, ev1 = timed(step_type1, steps, a.warmup)
        st1 = stage_ms(ev1, ["set_points", "spread", "fft", "deconv"])
        use_gather[0] = False                              # the type-2 region has no gather
        dt2, ev2 = timed(step_type2, steps, a.warmup)
        st2 = stage_ms(ev2, ["set_points", "deconv_pad", "fft", "interp"])
        exec1_ms = st1["spread"] + st1["fft"] + st1["deconv"]
        exec2_ms = st2["deconv_pad"] + st2["fft"] + st2["interp"]
        jobs = 1 if P.get("sharded") else world
        engine_used = plan.spread_engine_used()          # the per-point-set decision read back from the device (after the timed regions)
        # Halo variant of the spreading ring: the FFT stage's first pass adds the side buffer of the stencil reach, i.e. part of the
        # "spread" work is timed under "fft".  How much: the same FFT stage behind the self-contained nufft_spread (plain first pass).
        fft_plain_ms = None
        if int(info.ring_halo) and engine_used == "marching_ring":
            s = stream_ptr()
            ts = []
            for _ in range(6):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                _check(lib.nufft_spread(plan._handle, _ptr_table(vps), s))
                e0.record()
                _check(lib.nufft_fft_forward(plan._handle, s))
                e1.record()
                e1.synchronize()
                ts.append(e0.elapsed_time(e1))
            fft_plain_ms = float(np.mean(ts[1:]))
        rec = {
            "evalmode": "Direct" if evalmode_name == "direct" else "FastApproximation",
            # independent problems: every rank transforms its own Np points; sharded components: the job is ONE transform of Np points
            "value": jobs * Np * steps / dt1, "ms_per_step": dt1 / steps * 1e3,
            "type1": {"stages_ms": st1, "exec_only_pts_per_s": Np / (exec1_ms * 1e-3), "with_set_points_pts_per_s": jobs * Np * steps / dt1},
            "type2": {"stages_ms": st2, "exec_only_pts_per_s": Np / (exec2_ms * 1e-3), "with_set_points_pts_per_s": jobs * Np * steps / dt2,
                      "ms_per_step": dt2 / steps * 1e3},
            "workspace_bytes": int(plan.info().workspace_bytes),       # plan-owned device memory with this point set in place
            "ring_column": [int(info.ring_column[0]), int(info.ring_column[1])], "ring_segments": int(info.ring_segments),
            "ring_halo": int(info.ring_halo), "fft_plain_ms": fft_plain_ms,
            "sort_columns": bool(plan.sort_columns_used()), "sort_method": plan.sort_method_used(), "sort_column_bins": [int(info.sort_column[0]), int(info.sort_column[1])],
            "spread_engine": engine_used, "patch_f32acc": int(info.patch_f32acc), "patch_dims": [int(info.patch_dims[0]), int(info.patch_dims[1])], "patch_planar": int(info.patch_planar),
            "oversampled": [int(x) for x in plan.oversampled_dims], "size": [int(x) for x in plan.size],
            "spread_tile": [int(info.spread_tile[d]) for d in range(3)], "interp_tile": [int(info.interp_tile[d]) for d in range(3)],
        }
        del plan, uhat, vout
        torch.cuda.empty_cache()
        return rec

    P = prepare(cfg)
    Np, Cn, is_complex, real_bytes = P["Np"], P["Cn"], P["is_complex"], P["real_bytes"]
    P_sharded, C_total = bool(P.get("sharded")), P["C_total"]
    head = measure(P, a.evalmode, steps, True)
    other = measure(P, "fast" if a.evalmode == "direct" else "direct", steps, True) if full else None

    ab = algorithmic_bytes(Np, head["oversampled"], head["size"], is_complex, real_bytes, Cn)
    st1, st2 = head["type1"]["stages_ms"], head["type2"]["stages_ms"]
    exec1_ms = st1["spread"] + st1["fft"] + st1["deconv"]
    exec2_ms = st2["deconv_pad"] + st2["fft"] + st2["interp"]
    # the dominant kernel of the step: spreading (one launch per component)
    tname = {"float64": "double", "float32": "float", "complex64": "float", "complex128": "double"}[cfg["Z"]]
    if head["spread_engine"] == "mfma_patches" and head["patch_f32acc"]:
        kname = f"spread_patch32_kernel<{cfg['m']}"
        binding = ("the FP32 matrix pipe (v_mfma_f32_16x16x4, about half of the kernel) plus the vector work that shares its ALUs "
                   "(window evaluation, operand products) at one wave per SIMD, not HBM: see DESIGN.md section 4.5")
    elif head["spread_engine"] == "mfma_patches":
        kname = f"spread_patch_kernel<{tname}, {'true' if is_complex else 'false'}, {cfg['m']}, false, {head['patch_planar']}>"
        binding = ("issue and latency of the per-visit point set-up at 8 waves per CU (VALU 36 %, LDS 44 %, FP64 matrix pipe 16 % "
                   "busy), not HBM: see DESIGN.md section 4.4")
    elif head["spread_engine"] == "marching_ring_dense":
        kname = f"spread_march_dense_kernel<{tname}, {cfg['m']}, {'false' if head['evalmode'] == 'Direct' else 'true'}>"
        binding = ("the FP64 matrix pipe (v_mfma_f64_16x16x4: the points of a 4^3-cell bin accumulated in registers, NT instructions per four points) "
                   "and the vector work that builds its operands, then one flush of 4 NT LDS atomics per bin; not HBM: see DESIGN.md section 4.12")
    elif head["spread_engine"] == "marching_ring":
        kname = f"spread_march_kernel<{tname}, {'true' if is_complex else 'false'}, {cfg['m']}, {'false' if head['evalmode'] == 'Direct' else 'true'},"
        binding = ("the LDS atomic pipe (ds_add_f64: 8 array cycles per 64-lane wave instruction, 11.9 of them per point at 1.49 visits; "
                   "LDS array 73 % busy), then the two barriers per bin layer; not HBM: see DESIGN.md section 4.9")
        if head.get("ring_halo"):
            binding = ("the LDS atomic pipe (ds_add_f64: 8 array cycles per 64-lane wave instruction, 8 per point: halo variant, every point "
                       "spread once by its own column), then the two barriers per bin layer; the stencil reach leaves through a side buffer "
                       "that the FFT stage's first pass adds (frac_incl_consumer charges that pass's extra time to this stage); not HBM: "
                       "see DESIGN.md section 4.9")
    else:
        kname = f"spread_tile_kernel<{tname}, {'true' if is_complex else 'false'}, 3, {cfg['m']}"
        binding = ("LDS float atomics (ds_add_f64, 8.5 cycles per wave instruction per CU) and the scalar/vector issue of the "
                   "clipped stencil loop, not HBM: see DESIGN.md section 4.2")
    spread_s = st1["spread"] * 1e-3
    # halo variant: the side buffer (0.49 G at 32 x 32 columns, m = 4) is written by the kernel on top of G + P, and the time its
    # consumer adds to the FFT stage belongs to the zero + spread stage of SURVEY 8(d)
    halo_extra_s = max(0.0, (st1["fft"] - head["fft_plain_ms"]) * 1e-3) if head.get("fft_plain_ms") else 0.0
    if head.get("ring_halo"):
        n1c, n2c, mm = head["ring_column"][0], head["ring_column"][1], cfg["m"]
        xr, yr = (mm - 1) + ((mm - 1) & 1) + mm, 2 * mm - 1
        ab["spread_kernel_min"] += Cn * ab["G"] * ((n1c + xr) * (n2c + yr) / float(n1c * n2c) - 1.0)
    traffic_b, traffic_src, profile_us = pmc_traffic(kname, a.config, head["evalmode"])
    sq = sq_counters(kname, a.config, head["evalmode"])
    # (the interpolation ring's name carries the window mode as its last template argument; the lookup is restricted to this mode's profiles as well)
    interp_kname = (("interp_march_staged_kernel" if head.get("sort_columns") else "interp_march_kernel")
                    + f"<{tname}, {'true' if is_complex and cfg['Z'] != 'complex128' else 'false'}, {cfg['m']}, {'false' if head['evalmode'] == 'Direct' else 'true'}>")
    interp_traffic_b, interp_traffic_src, interp_profile_us = pmc_traffic(interp_kname, a.config, head["evalmode"])
    probe = hbm_probe(dev) if full else None
    peak_m = probe["peak_measured_GBs"] if probe else None
    achieved = ab["spread_kernel"] / spread_s / 1e9
    roofline = {
        "bound": "hbm",
        "kernel": kname + ("" if kname.endswith(">") else ", ...>") + " (the zero + spread stage of the C = %d component(s))" % Cn,
        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
        "peak_measured": peak_m, "frac_of_measured_peak": (achieved / peak_m) if peak_m else None,
        "traffic": (traffic_b / 1e9) if traffic_b is not None else None,
        "traffic_unit": "GB per launch = per stage (PMC FETCH_SIZE x2 + WRITE_SIZE; one launch covers the C components)", "traffic_source": traffic_src,
        # what binds the kernel: the design note, and — where a counter profile of this configuration is committed — the
        # SQ counters it rests on, read from profiles/*_sq.json (not typed in here)
        "binding_resource": binding,
        "binding_resource_counters": sq if sq is not None else "no SQ counter profile of this configuration is committed (scripts/profile_bench.sh)",
        # the profile the traffic figure comes from: its average duration of this kernel, and whether this run agrees with it
        "profile_kernel_us": profile_us,
        "profile_matches_run": (abs(profile_us / 1e3 - st1["spread"] / (Cn if head["spread_engine"] == "mfma_patches" and not head["patch_planar"] else 1)) < 0.1 * st1["spread"]) if profile_us else None,
        "algorithmic_bytes_per_stage": ab["spread_kernel"],
        "algorithmic_bytes_note": "SURVEY 8(d): zero + spread = 3G + P per component (what the reference's algorithm moves); the "
                                  "kernel's own compulsory traffic is G + P (every cell written once, no zero fill): achieved_own_traffic",
        "own_traffic_bytes_per_stage": ab["spread_kernel_min"],
        "achieved_own_traffic": ab["spread_kernel_min"] / spread_s / 1e9,
        "frac_own_traffic": ab["spread_kernel_min"] / spread_s / 1e9 / HBM_PEAK_GBS,
        "frac_own": ab["spread_kernel_min"] / spread_s / 1e9 / HBM_PEAK_GBS,      # the kernel's own compulsory traffic (G + P [+ side buffer]) against the peak
        "kernel_ms": st1["spread"],
        # halo variant only: the stage charged with the extra time of the FFT pass that adds the side buffer (else = frac)
        "consumer_extra_ms": halo_extra_s * 1e3,
        "frac_incl_consumer": ab["spread_kernel"] / (spread_s + halo_extra_s) / 1e9 / HBM_PEAK_GBS,
        "interp": {"kernel_ms": st2["interp"], "algorithmic_bytes_per_stage": ab["interp_kernel"],
                   "achieved": ab["interp_kernel"] / (st2["interp"] * 1e-3) / 1e9, "frac": ab["interp_kernel"] / (st2["interp"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                   "kernel": interp_kname,
                   "traffic": (interp_traffic_b / 1e9) if interp_traffic_b is not None else None, "traffic_source": interp_traffic_src,
                   "profile_kernel_us": interp_profile_us,
                   # the interpolation stage is one launch of this kernel (+ a stand-by launch of the tile kernel that exits on the device flag)
                   "profile_matches_run": (abs(interp_profile_us / 1e3 - st2["interp"]) < 0.1 * st2["interp"]) if interp_profile_us else None,
                   "binding_resource_counters": sq_counters(interp_kname, a.config, head["evalmode"]),
                   "note": "type-2 gather stage (R(G) + R(points) + W(values)); z-marching LDS ring where the point set is not sliced"},
        "type1_exec_frac": ab["type1_exec"] / (exec1_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "type2_exec_frac": ab["type2_exec"] / (exec2_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "type1_exec_frac_of_measured_peak": (ab["type1_exec"] / (exec1_ms * 1e-3) / 1e9 / peak_m) if peak_m else None,
        "type2_exec_frac_of_measured_peak": (ab["type2_exec"] / (exec2_ms * 1e-3) / 1e9 / peak_m) if peak_m else None,
        "hbm_probe": probe,
    }
    tname_z = {"float64": "Float64", "float32": "Float32", "complex64": "ComplexF32", "complex128": "ComplexF64"}[cfg["Z"]]
    result = {
        "metric": f"NU-points/s, type-1 NUFFT (set_points! + exec_type1!), {cfg['n']}^3 {tname_z} m={cfg['m']}, {head['evalmode']}() window"
                  + (" (the reference's ROCBackend default)" if head["evalmode"] == "Direct" else ""),
        "value": head["value"],
        "unit": "NU-points/s",
        "n_gpus": world,
        **({"shared_gpu_self_test": True} if SHARE_GPU_SELF_TEST else {}),
        "steps": steps,
        "warmup": a.warmup,
        "ms_per_step": head["ms_per_step"],
        "higher_is_better": True,
        "scaling": "strong" if P_sharded else "weak",
        "vs_baseline": None,
        "dtype": {"float64": "f64", "float32": "f32", "complex64": "c64 (f32 arithmetic and accumulation, as the reference)", "complex128": "c128"}[cfg["Z"]],
        "data": "synthetic",
        "config": {
            "workload": f"{cfg['label']}, m={cfg['m']}, sigma={cfg['sigma']} (oversampled {tuple(head['oversampled'])}), "
                        f"ntransforms={Cn}, {head['evalmode']} window (the other evaluation mode: sibling record)",
            "protocol": "set_points! + exec_type1! per step, inputs resident in HBM (reference benchmark protocol)",
            "spread_engine": head["spread_engine"], "spread_tile": head["spread_tile"], "interp_tile": head["interp_tile"],
            "ring_column": head["ring_column"], "ring_segments": head["ring_segments"], "ring_halo": head["ring_halo"], "workspace_bytes": head["workspace_bytes"],
            "sort_columns": head["sort_columns"], "sort_method": head["sort_method"], "sort_column_bins": head["sort_column_bins"],
            "set_points_ms": st1["set_points"], "spread_ms": st1["spread"], "fft_ms": st1["fft"], "deconv_ms": st1["deconv"], "interp_ms": st2["interp"],
            "type2_value": head["type2"]["with_set_points_pts_per_s"],
            "parallelism": (f"ntransforms = {C_total} components of one transform sharded over {world} GPU(s) (component c on rank c mod N, same points)"
                            if P_sharded else f"{world} independent plan(s), one per GPU")
                           + ("" if not distributed or a.no_gather else "; RCCL gather of spectra to rank 0 overlapped on a side stream"),
        },
        "roofline": roofline,
        "type1": head["type1"], "type2": head["type2"],
    }
    if other is not None:
        key = "fast_approximation" if other["evalmode"] == "FastApproximation" else "direct"
        o_spread_s = other["type1"]["stages_ms"]["spread"] * 1e-3
        result[key] = {"evalmode": other["evalmode"], "value": other["value"], "ms_per_step": other["ms_per_step"],
                       "type1": other["type1"], "type2": other["type2"], "spread_engine": other["spread_engine"],
                       "roofline_frac": ab["spread_kernel"] / o_spread_s / 1e9 / HBM_PEAK_GBS,
                       "roofline_frac_of_measured_peak": (ab["spread_kernel"] / o_spread_s / 1e9 / peak_m) if peak_m else None}
    # The ROC default window (Direct) as a scalar inside `config`, and the other single-GPU BASELINE configurations (C3, C4)
    # as compact records there too, so that they are part of the driver-run line and not only of builder-run profiles.
    direct_rec = head if head["evalmode"] == "Direct" else other
    if direct_rec is not None:
        result["config"]["direct_value"] = direct_rec["value"]
        result["config"]["direct_ms_per_step"] = direct_rec["ms_per_step"]
    fast_rec = head if head["evalmode"] == "FastApproximation" else other
    if fast_rec is not None:
        result["config"]["fast_value"] = fast_rec["value"]
        result["config"]["fast_ms_per_step"] = fast_rec["ms_per_step"]
        result["config"]["fast_type2_value"] = fast_rec["type2"]["with_set_points_pts_per_s"]
    if full and a.config == "c2" and not a.no_other_configs:
        del P
        torch.cuda.empty_cache()
        others = {}
        for name in ("c4", "c3"):      # (C4 first: the long C3 run leaves the device at a lower clock)
            try:
                oc = dict(CONFIGS[name])
                Po = prepare(oc)
                # (polynomial window as in rounds 1-4, so that these records stay comparable; the Direct() value beside it)
                nst = 5 if name == "c4" else 3
                r = measure(Po, "fast", nst, False)
                rd = measure(Po, "direct", nst, False)
                abo = algorithmic_bytes(Po["Np"], r["oversampled"], r["size"], Po["is_complex"], Po["real_bytes"], Po["Cn"])
                sp_ms, ip_ms = r["type1"]["stages_ms"]["spread"], r["type2"]["stages_ms"]["interp"]
                others[name] = {
                    "workload": oc["label"] + f", m={oc['m']}, sigma={oc['sigma']}, {r['evalmode']} window",
                    "value": r["value"], "ms_per_step": r["ms_per_step"], "steps": nst,
                    "type2_value": r["type2"]["with_set_points_pts_per_s"], "type2_ms_per_step": r["type2"]["ms_per_step"],
                    "spread_ms": sp_ms, "interp_ms": ip_ms, "spread_engine": r["spread_engine"], "ring_halo": r["ring_halo"],
                    "fft_ms": r["type1"]["stages_ms"]["fft"], "fft_plain_ms": r["fft_plain_ms"],
                    "roofline_frac": abo["spread_kernel"] / (sp_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "roofline_frac_own_traffic": abo["spread_kernel_min"] / (sp_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "interp_roofline_frac": abo["interp_kernel"] / (ip_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "workspace_bytes": r["workspace_bytes"],
                    "set_points_ms": r["type1"]["stages_ms"]["set_points"], "sort_columns": r["sort_columns"], "sort_method": r["sort_method"],
                    "direct_value": rd["value"], "direct_type2_value": rd["type2"]["with_set_points_pts_per_s"],
                    "direct_spread_ms": rd["type1"]["stages_ms"]["spread"], "direct_interp_ms": rd["type2"]["stages_ms"]["interp"],
                }
                if name == "c3":
                    # C3 is bound by arithmetic, not HBM: (2M)^3 multiply-adds per point and component on the FP32 pipes
                    flop = 2.0 * (2 * oc["m"]) ** 3 * (2 if Po["is_complex"] else 1) * Po["Np"] * Po["Cn"]
                    others[name]["roofline_fp32"] = {"bound": "fp32", "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                                     "achieved_tflops": flop / (sp_ms * 1e-3) / 1e12, "frac": flop / (sp_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS,
                                                     "flop_per_stage": flop, "kernel": "spread_patch32_kernel (v_mfma_f32_16x16x4 + vector FP32)",
                                                     "interp_achieved_tflops": flop / (ip_ms * 1e-3) / 1e12,
                                                     "interp_frac": flop / (ip_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS}
                    others[name]["fp32_achieved_tflops"] = others[name]["roofline_fp32"]["achieved_tflops"]
                    others[name]["fp32_frac"] = others[name]["roofline_fp32"]["frac"]
                    others[name]["interp_fp32_achieved_tflops"] = others[name]["roofline_fp32"]["interp_achieved_tflops"]
                    others[name]["interp_fp32_frac"] = others[name]["roofline_fp32"]["interp_frac"]
                del Po
                torch.cuda.empty_cache()
            except Exception as exc:           # informative records: never lose the headline line over them
                others[name] = {"error": repr(exc)}
        result["config"]["other_configs"] = others
        # ... and flattened into scalars (the driver's parser keeps scalars of `config`, not nested records)
        for name, r in others.items():
            if "error" in r:
                result["config"][f"{name}_error"] = r["error"]
                continue
            for k in ("value", "ms_per_step", "type2_value", "type2_ms_per_step", "spread_ms", "interp_ms", "spread_engine", "roofline_frac",
                      "ring_halo", "fft_ms", "fft_plain_ms", "set_points_ms", "sort_columns", "sort_method", "direct_value", "direct_type2_value", "direct_spread_ms", "direct_interp_ms",
                      "roofline_frac_own_traffic", "interp_roofline_frac", "workspace_bytes", "fp32_achieved_tflops", "fp32_frac",
                      "interp_fp32_achieved_tflops", "interp_fp32_frac"):
                if k in r:
                    result["config"][f"{name}_{k}"] = r[k]
    if full and a.config == "c2" and not a.no_reference_protocol:
        result["reference_protocol"] = reference_protocol(cfg, nufft, dev, sweep=not a.no_density_sweep)
        torch.cuda.empty_cache()
        try:       # the reference's default element type, next to its published ComplexF64 table (tests/golden/reference_dat.json: ComplexF64_ROC_shared)
            result["reference_protocol_complexf64"] = reference_protocol(cfg, nufft, dev, sweep=not a.no_density_sweep, complex_data=True)
            rp = result["reference_protocol_complexf64"]
            result["config"]["refproto_c128_type1_value"] = rp["type1_pts_per_s"]
            result["config"]["refproto_c128_type2_value"] = rp["type2_pts_per_s"]
        except Exception as exc:
            result["reference_protocol_complexf64"] = {"error": repr(exc)}
        rp = result["reference_protocol"]
        result["config"]["refproto_f64_type1_value"] = rp["type1_pts_per_s"]
        result["config"]["refproto_f64_type2_value"] = rp["type2_pts_per_s"]
    if rank == 0 and full and not a.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(cfg)
    result["config"] = lead_config(result["config"])
    if rank == 0:
        print(json.dumps(result), flush=True)
    if distributed:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def published_rows():
    try:
        return json.load(open(os.path.join(ROOT, "tests", "golden", "reference_dat.json")))["sets"]
    except Exception:
        return {}


def reference_protocol(cfg, nufft, dev, sweep=True, complex_data=False):
    """The reference's published benchmark protocol, for comparison with BASELINE.md (not the headline metric):
    sigma = 1.5, BackwardsKaiserBessel with Direct() evaluation (the ROC defaults), coordinates ~ N(0, 1) folded
    into the period, time = set_points! + exec! + sync, median over repetitions
    (benchmark/CPU+AMDGPU/run_benchmarks.jl:57-90), and its density sweep rho = Np / N^3 in 10^(-4:0.5:1) (:287-291)
    next to the rows the reference published for MI300A (tests/golden/reference_dat.json)."""
    n, m = cfg["n"], cfg["m"]
    dims = (n, n, n)
    Zt = torch.complex128 if complex_data else torch.float64      # (ComplexF64 is the reference's default element type)
    plan = nufft.PlanNUFFT(Zt, dims, m=m, sigma=1.5, kernel_evalmode=nufft.Direct(), backend=nufft.ROCBackend(dev.index or 0))
    u = torch.empty(plan.shape, dtype=torch.complex128, device=dev)
    pub = {r["Np"]: r for r in published_rows().get("ComplexF64_ROC_shared" if complex_data else "Float64_ROC_shared", {}).get("rows", [])}

    def run(Np, reps):
        g = torch.Generator(device=dev).manual_seed(4242)
        xs = tuple(torch.randn(Np, dtype=torch.float64, device=dev, generator=g) for _ in dims)
        v = torch.randn(Np, dtype=torch.float64, device=dev, generator=g)
        if complex_data:
            v = torch.complex(v, torch.randn(Np, dtype=torch.float64, device=dev, generator=g))
        out = torch.empty(Np, dtype=Zt, device=dev)
        res = {}
        for name, fn in (("type1", lambda: nufft.exec_type1(u, plan, v)), ("type2", lambda: nufft.exec_type2(out, plan, u))):
            nufft.set_points(plan, xs); fn()
            ts = []
            for _ in range(reps):
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                nufft.set_points(plan, xs); fn()
                torch.cuda.synchronize(dev)
                ts.append(time.perf_counter() - t0)
            res[name + "_s"] = float(np.median(ts))
            res[name + "_pts_per_s"] = Np / res[name + "_s"]
        return res

    Np = int(cfg["np"])
    res = run(Np, 10)
    res["config"] = (f"Ns={n}^3, Np={Np:.0e} ~ N(0,1) folded, {'ComplexF64' if complex_data else 'Float64'}, m={m}, sigma=1.5 (oversampled {plan.oversampled_dims}), "
                     "Direct window, set_points! + exec! + sync, median of 10")
    res["published_mi300a_pts_per_s"] = ({"type1": "1.88e8-2.50e8", "type2": "3.83e8-6.38e8", "source": "BASELINE.md (Np = 1.7e7 ... 1.7e8)"} if complex_data else
                                         {"type1": "2.4e8-2.7e8", "type2": "6.2e8-9.6e8", "source": "BASELINE.md (Np = 1.7e7 ... 1.7e8)"})
    res["spread_engine"] = plan.spread_engine_used()
    if sweep:
        rows = []
        for k in (range(6, 11) if complex_data else range(11)):      # rho = 10^(-4 + k / 2), Np = round(rho N^3): the reference's list (complex data: rho >= 0.1)
            rho = 10.0 ** (-4 + 0.5 * k)
            Npk = int(round(rho * n ** 3))
            r = run(Npk, 5 if Npk > 2e7 else 8)
            row = {"rho": rho, "Np": Npk, "type1_s": r["type1_s"], "type2_s": r["type2_s"],
                   "type1_pts_per_s": r["type1_pts_per_s"], "type2_pts_per_s": r["type2_pts_per_s"]}
            if Npk in pub:
                row["mi300a_published"] = {"type1_s": pub[Npk]["type1_s"], "type2_s": pub[Npk]["type2_s"]}
            rows.append(row)
        res["density_sweep"] = rows
    return res


def cpu_baseline(cfg):
    """The oracle's C restatement of the reference's blocked CPU algorithm (+ pocketfft), timed on the
    host cores of this box on a bounded sample of the same workload (kind = "port": the reference's
    Julia CPU backend cannot run here — no Julia runtime)."""
    note = {}
    rows = published_rows().get("Float64_CPU", {})
    if rows:
        big = [r for r in rows["rows"] if r["Np"] >= 1.6e7][:1]
        if big:
            note = {"reference_published_cpu": {"pts_per_s_type1": big[0]["Np"] / big[0]["type1_s"], "Np": big[0]["Np"],
                                                "device": rows["header"].get("Device"), "sigma": 1.5,
                                                "source": "benchmark/CPU+AMDGPU/results.MI300A_adastra/" + rows["file"],
                                                "note": "the reference's own CPU backend on its machine (other hardware, sigma = 1.5, denser "
                                                        "points); this port merges 13.8k-cell padded blocks holding ~300 points each at "
                                                        "C2's density"}}
    try:
        from oracle import c_oracle as CO, nufft_oracle as O
        if cfg["Z"] != "float64" or cfg["C"] != 1:
            return dict({"value": None, "unit": "NU-points/s", "cores": 0, "kind": "port",
                         "sample": "timed on the C2 workload only (bench.py --config c2)"}, **note)
        if not CO.available():
            return dict({"value": None, "unit": "NU-points/s", "cores": 0, "kind": "port", "sample": "oracle/libnufft_oracle.so not built"}, **note)
        cores = CO.num_threads()
        n, m, sigma, Np_full = cfg["n"], cfg["m"], cfg["sigma"], int(cfg["np"])
        dims = (n, n, n)
        oplan = O.OraclePlan(dims, is_real=True, M=m, sigma=sigma, evalmode=O.FAST_APPROXIMATION)
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
        Np_s = int(min(Np_full, max(1_000_000, (12.0 - t_probe) / per_pt)))
        t = run(Np_s)
        return dict({"value": Np_s / t, "unit": "NU-points/s", "cores": cores, "kind": "port",
                     "sample": f"one set_points+type-1 transform, same grid ({n}^3, sigma={sigma}, m={m}), "
                               f"Np={Np_s} of {Np_full} points, {t:.1f} s wall; C/OpenMP blocked spreading (plain adds under a lock, "
                               f"the reference's default) + scipy pocketfft + C/OpenMP truncation and deconvolution, plan-owned work arrays "
                               f"(third transform of the plan), polynomial window (the reference's CPU default); threads = CPUs "
                               f"available to the process (affinity capped by the cgroup quota: {CO.available_cpus()} of {os.cpu_count()} logical CPUs)",
                     "seconds": t}, **note)
    except Exception as exc:  # the baseline is informative only
        return dict({"value": None, "unit": "NU-points/s", "cores": 0, "kind": "port", "sample": f"failed: {exc!r}"}, **note)


if __name__ == "__main__":
    main()
