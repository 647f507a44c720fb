"""CPU oracle probe (development helper): C blocked spreading, lock vs atomics merge, at the quota's thread count."""
import time, os, sys, subprocess
if len(sys.argv) > 1:
    import numpy as np
    sys.path.insert(0, os.getcwd())
    from oracle import c_oracle as CO, nufft_oracle as O
    plan = O.OraclePlan((256, 256, 256), is_real=True, M=4, sigma=2.0, evalmode=O.FAST_APPROXIMATION)
    rng = np.random.default_rng(1)
    for Np in (1_000_000, 8_000_000):
        xs = [rng.random(Np) * O.TWO_PI for _ in range(3)]; v = rng.standard_normal(Np)
        O.set_points(plan, xs)
        t = time.perf_counter(); CO.spread(plan, [v]); dt = time.perf_counter() - t
        t = time.perf_counter(); CO.exec_type1(plan, v); dt2 = time.perf_counter() - t
        print(f"atomics={os.environ.get('ORACLE_USE_ATOMICS','0')} threads {CO.num_threads():3d} Np {Np:.0e}: spread {dt:.3f} s, exec_type1 {dt2:.3f} s -> {Np / dt2 / 1e6:.2f} Mpts/s", flush=True)
else:
    for ua in ("0", "1"):
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, ORACLE_USE_ATOMICS=ua))
