"""Development helper: one point spread by the Float32 patch kernel vs the LDS tiles, cell by cell (stage-level grids)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nufft_pkg import nufft
M = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dims = (48, 48, 48)
ps = {e: nufft.PlanNUFFT(torch.complex64, dims, m=M, sigma=2.0, spread_method=e, kernel_evalmode=nufft.FastApproximation(), backend=nufft.ROCBackend(0)) for e in ("lds_tiles", "mfma_patches")}
print("f32acc", ps["mfma_patches"].info().patch_f32acc, "patch rows", ps["mfma_patches"].info().patch_dims[1])
No = ps["lds_tiles"].oversampled_dims
for cell in ((50, 41, 30), (50, 41, 26), (50, 45, 30)):
    xs = tuple(torch.tensor([(cell[k] + 0.3) / No[k] * 2 * np.pi], dtype=torch.float32, device="cuda") for k in range(3))
    v = torch.tensor([1.0 + 2.0j], dtype=torch.complex64, device="cuda")
    g = {}
    for e in ps:
        nufft.set_points(ps[e], xs)
        nufft.spread_from_points(ps[e], v)
        g[e] = nufft.oversampled_grid(ps[e]).cpu().numpy().copy()
    a, b = g["lds_tiles"], g["mfma_patches"]          # shape (z, y, x)
    print("cell", cell, "norms", np.linalg.norm(a), np.linalg.norm(b), "rel diff", np.linalg.norm(a - b) / np.linalg.norm(a))
    nza, nzb = np.argwhere(np.abs(a) > 0), np.argwhere(np.abs(b) > 0)
    print("  tiles nonzero box z,y,x:", nza.min(0), nza.max(0), " patches:", (nzb.min(0), nzb.max(0)) if len(nzb) else None)
    # per-plane comparison around the stencil
    z0, y0, x0 = nza.min(0)
    for z in range(z0, nza.max(0)[0] + 1):
        pa, pb = a[z], b[z]
        print(f"   z={z}: |a|={np.linalg.norm(pa):.4e} |b|={np.linalg.norm(pb):.4e} diff={np.linalg.norm(pa - pb):.3e}")
    # the x row through the stencil centre
    zc, yc = (nza.min(0)[0] + nza.max(0)[0]) // 2, (nza.min(0)[1] + nza.max(0)[1]) // 2
    print("   row a:", np.round(a[zc, yc, x0:x0 + 2 * M], 4))
    print("   row b:", np.round(b[zc, yc, x0:x0 + 2 * M], 4))
    print("   col a (y):", np.round(a[zc, y0:y0 + 2 * M, x0 + M], 4))
    print("   col b (y):", np.round(b[zc, y0:y0 + 2 * M, x0 + M], 4))

    xc = x0 + M - 1
    for z in range(z0 - 1, nza.max(0)[0] + 2):
        print(f"   z={z} y-profile at x={xc}: a", np.round(a[z % No[2], y0 - 8:y0 + 2 * M + 4, xc].real, 4))
        print(f"   z={z} y-profile at x={xc}: b", np.round(b[z % No[2], y0 - 8:y0 + 2 * M + 4, xc].real, 4))
