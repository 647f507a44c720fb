"""Development helper: single-point type-1 transforms of the LDS-tile engine, cube accumulation on vs off."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nufft_pkg import nufft
dims = (64, 64, 64)
ps = {}
for c in ("0", "1"):
    os.environ["NUFFT_SPREAD_CUBES"] = c
    ps[c] = nufft.PlanNUFFT(torch.float64, dims, m=4, sigma=2.0, spread_method="lds_tiles", kernel_evalmode=nufft.FastApproximation(), backend=nufft.ROCBackend(0))
No = ps["0"].oversampled_dims
print("oversampled", No, "tile", list(ps["0"].info().spread_tile))
bad = []
for d in range(3):
    for c in range(0, No[d], 1):
        cell = [No[0] // 2 + 1, No[1] // 2 + 1, No[2] // 2 + 1]
        cell[d] = c
        xs = tuple(torch.tensor([(cell[k] + 0.3) / No[k] * 2 * np.pi], dtype=torch.float64, device="cuda") for k in range(3))
        v = torch.ones(1, dtype=torch.float64, device="cuda")
        outs = []
        for e in ("0", "1"):
            nufft.set_points(ps[e], xs)
            u = torch.empty(ps[e].shape, dtype=torch.complex128, device="cuda")
            nufft.exec_type1(u, ps[e], v)
            outs.append(u)
        r = float((outs[0] - outs[1]).norm() / outs[0].norm())
        if r > 1e-12:
            bad.append((d, c, "%.2e" % r))
print("bad positions (dim, cell, rel):", bad[:60], len(bad))
g = torch.Generator(device="cuda").manual_seed(5)
for Np in (2, 3, 4, 5, 8, 9, 16, 64, 500):
    for trial in range(3):
        # points confined to a small box so that they share tiles / runs / chunks
        base = torch.tensor([1.0, 2.0, 3.0], dtype=torch.float64, device="cuda")
        xs = tuple((base[k] + torch.rand(Np, dtype=torch.float64, device="cuda", generator=g) * 0.25).contiguous() for k in range(3))
        v = torch.randn(Np, dtype=torch.float64, device="cuda", generator=g)
        outs = []
        for e in ("0", "1"):
            nufft.set_points(ps[e], xs)
            u = torch.empty(ps[e].shape, dtype=torch.complex128, device="cuda")
            nufft.exec_type1(u, ps[e], v)
            outs.append(u)
        print(Np, trial, "rel %.2e" % float((outs[0] - outs[1]).norm() / outs[0].norm()))
print("--- n points in ONE cell")
for Np in (1, 2, 4, 5, 6, 8, 9, 12, 17):
    xs = tuple((torch.full((Np,), (70 + 0.3) / 128 * 2 * np.pi, dtype=torch.float64, device="cuda") + 1e-3 * torch.rand(Np, dtype=torch.float64, device="cuda", generator=g)).contiguous() for k in range(3))
    v = torch.randn(Np, dtype=torch.float64, device="cuda", generator=g)
    outs = []
    for e in ("0", "1"):
        nufft.set_points(ps[e], xs)
        u = torch.empty(ps[e].shape, dtype=torch.complex128, device="cuda")
        nufft.exec_type1(u, ps[e], v)
        outs.append(u)
    print(Np, "rel %.2e" % float((outs[0] - outs[1]).norm() / outs[0].norm()))
