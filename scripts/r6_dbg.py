"""Dense-window engine against the atomic window on single points / small sets: where do the grids differ?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nufft_pkg import nufft
Z = {"f64": np.float64, "f32": np.float32}[sys.argv[1] if len(sys.argv) > 1 else "f64"]
M = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dims = (48, 48, 56)
T = Z
def mk(opts):
    return nufft.PlanNUFFT(Z, dims, m=M, sigma=2.0, kernel_evalmode=nufft.Direct(), spread_method="marching_ring", backend=nufft.ROCBackend(0),
                           options=dict({"NUFFT_SMARCH_HALO": 2}, **opts))
pa, pb = mk({"NUFFT_DENSE": 0}), mk({"NUFFT_DENSE_MIN": 0})
print("ring column", list(pa.info().ring_column), "halo", pa.info().ring_halo, "Nover", pa.oversampled_dims)
rng = np.random.default_rng(3)
for name, Np in (("one", 1), ("two", 2), ("five", 5), ("many", 20000)):
    xs = [(rng.random(Np) * 2 * np.pi).astype(T) for _ in dims]
    if name == "one":
        xs = [np.array([1.0], dtype=T), np.array([2.0], dtype=T), np.array([3.0], dtype=T)]
    v = rng.standard_normal(Np).astype(Z) if Np > 1 else np.ones(1, dtype=Z)
    gs = []
    for p in (pa, pb):
        nufft.set_points(p, tuple(torch.from_numpy(x).cuda() for x in xs))
        nufft.spread_from_points(p, torch.from_numpy(v).cuda())
        gs.append(nufft.oversampled_grid(p, 0).cpu().numpy().astype(np.float64))
        print("  engine", p.spread_engine_used(), "sort", p.sort_method_used())
    ga, gb = gs
    d = np.abs(ga - gb)
    print(name, "rel", np.linalg.norm(ga - gb) / np.linalg.norm(ga), "norms", np.linalg.norm(ga), np.linalg.norm(gb), "nnz", (ga != 0).sum(), (gb != 0).sum())
    if name == "one":
        ia = np.argwhere(ga != 0); ib = np.argwhere(gb != 0)
        print("  atomic window support z,y,x:", ia.min(0), ia.max(0), " dense:", ib.min(0) if len(ib) else None, ib.max(0) if len(ib) else None)
        k = np.unravel_index(np.argmax(d), d.shape)
        print("  worst at", k, ga[k], gb[k])
        z0, y0, x0 = ia.min(0)
        print("  ratio sample", (gb[z0:z0+3, y0:y0+3, x0:x0+3] / ga[z0:z0+3, y0:y0+3, x0:x0+3]))
