"""Development helper: type-1 results of the two spreading engines on the same dense point set (cuda:0)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nufft_pkg import nufft

def run(Z, n, M, Np, C=1, sigma=2.0, dist="uniform"):
    T = torch.float32 if Z in (torch.float32, torch.complex64) else torch.float64
    g = torch.Generator(device="cuda").manual_seed(3)
    if dist == "uniform":
        xs = tuple(torch.rand(Np, dtype=T, device="cuda", generator=g) * (2 * np.pi) for _ in range(3))
    else:
        xs = tuple(torch.randn(Np, dtype=T, device="cuda", generator=g) * 0.3 + np.pi for _ in range(3))
    vs = tuple(torch.randn(Np, dtype=Z, device="cuda", generator=g) for _ in range(C))
    outs = []
    for eng in ("lds_tiles", "mfma_patches"):
        p = nufft.PlanNUFFT(Z, (n, n, n), m=M, sigma=sigma, ntransforms=C, spread_method=eng, kernel_evalmode=nufft.FastApproximation(), backend=nufft.ROCBackend(0))
        nufft.set_points(p, xs)
        us = tuple(torch.empty(p.shape, dtype=p.eltype, device="cuda") for _ in range(C))
        nufft.exec_type1(us if C > 1 else us[0], p, vs if C > 1 else vs[0])
        outs.append(us)
    for c in range(C):
        a, b = outs[0][c], outs[1][c]
        print(f"{Z} n={n} M={M} Np={Np} C={C} {dist} comp {c}: finite {bool(torch.isfinite(torch.view_as_real(b)).all())} rel {float((a - b).norm() / a.norm()):.3e}", flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "density":
        for Np in (2000, 20000, 100000, 300000, 1000000):
            run(torch.float64, 64, 4, Np)
        sys.exit(0)

    run(torch.float64, 64, 4, 2_000_000)
    run(torch.float64, 64, 4, 200_000, dist="cluster")
    run(torch.complex64, 64, 8, 1_000_000)
    run(torch.complex64, 128, 8, 1_000_000)
    run(torch.complex64, 128, 8, 10_000_000)
    run(torch.complex128, 64, 6, 2_000_000, C=2)
    run(torch.float32, 96, 5, 3_000_000)
    run(torch.complex64, 512, 8, 20_000_000)
