"""Development helper: single-point type-1 transforms, MFMA patches vs LDS tiles, over a lattice of positions."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nufft_pkg import nufft
dims = (36, 50, 40)
ps = {e: nufft.PlanNUFFT(torch.float64, dims, m=4, sigma=2.0, spread_method=e, backend=nufft.ROCBackend(0)) for e in ("lds_tiles", "mfma_patches")}
No = ps["lds_tiles"].oversampled_dims
print("oversampled", No)
bad = []
for d in range(3):
    for c in range(0, No[d], 1):
        cell = [No[0] // 2 + 1, No[1] // 2 + 1, No[2] // 2 + 1]
        cell[d] = c
        xs = tuple(torch.tensor([(cell[k] + 0.3) / No[k] * 2 * np.pi], dtype=torch.float64, device="cuda") for k in range(3))
        v = torch.ones(1, dtype=torch.float64, device="cuda")
        outs = []
        for e in ps:
            nufft.set_points(ps[e], xs)
            u = torch.empty(ps[e].shape, dtype=torch.complex128, device="cuda")
            nufft.exec_type1(u, ps[e], v)
            outs.append(u)
        r = float((outs[0] - outs[1]).norm() / outs[0].norm())
        if r > 1e-12:
            bad.append((d, c, r))
print("bad positions (dim, cell, rel):", bad[:40], len(bad))
