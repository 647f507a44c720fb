"""Prints, for the published Np rows of the reference's error tables, the ratio (HIP path error / published error) - 1 on
the published protocol with this box's own random inputs (tests/test_reference_dat.py holds the protocol)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import test_reference_dat as T  # noqa: E402
from nufft_pkg import nufft  # noqa: E402

for is_real in (True, False):
    for Np in [r["Np"] for r in T.GOLD["sets"]["Float64_CPU"]["rows"] if r["Np"] >= 1000000]:
        for mode, name in ((nufft.Direct(), "ROC_shared"), (nufft.FastApproximation(), "CPU")):
            setn = ("Float64_" if is_real else "ComplexF64_") + name
            r1, r2 = T.published(setn, Np)
            for seed in (1, 2):
                e1, e2 = T._gpu_errors(torch, nufft, Np, is_real, mode, seed=seed)
                print(f"{setn:24s} Np={Np:9d} seed={seed} e1/pub-1={e1 / r1 - 1:+.3e} e2/pub-1={e2 / r2 - 1:+.3e}", flush=True)
