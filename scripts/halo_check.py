"""Development check of the spreading ring's halo variant: type-1 outputs and stage-level grids of NUFFT_SMARCH_HALO=2 (fused and
unfused) against NUFFT_SMARCH_HALO=0, several shapes.  Each variant runs in its own process (the switches are latched at plan creation)."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [  # (n, sigma, m, Z, Np, C, mode)
    (64, 2.0, 4, "f64", 200000, 1, "poly"), (64, 2.0, 4, "f64", 200000, 1, "direct"), (64, 1.5, 4, "f64", 100000, 2, "poly"),
    (64, 2.0, 3, "f64", 100000, 1, "poly"), (64, 2.0, 5, "f32", 100000, 1, "poly"), (64, 2.0, 2, "f32", 100000, 3, "poly"),
    (96, 2.0, 6, "f64", 100000, 1, "poly"), (128, 2.0, 4, "f64", 1000000, 1, "poly"), (64, 2.0, 7, "f64", 50000, 1, "poly"),
    (128, 2.0, 8, "f32", 50000, 1, "direct"),
    (64, 2.0, 4, "c128", 100000, 1, "poly"), (64, 2.0, 3, "c64", 100000, 2, "direct"), (128, 2.0, 2, "c128", 100000, 1, "poly"),
    (64, 1.5, 4, "c64", 100000, 1, "poly"), (128, 2.0, 5, "c128", 50000, 1, "poly"),
]

if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    from nufft_pkg import nufft
    out = {}
    for ci, (n, sigma, m, z, Np, C, mode) in enumerate(CASES):
        Z = {"f64": torch.float64, "f32": torch.float32, "c128": torch.complex128, "c64": torch.complex64}[z]
        T = torch.float64 if z in ("f64", "c128") else torch.float32
        g = torch.Generator(device="cuda").manual_seed(100 + ci)
        xs = tuple(torch.rand(Np, dtype=T, device="cuda", generator=g) * (2 * np.pi) for _ in range(3))
        vs = tuple(torch.randn(Np, dtype=Z, device="cuda", generator=g) for _ in range(C))
        ev = nufft.Direct() if mode == "direct" else nufft.FastApproximation()
        p = nufft.PlanNUFFT(Z, (n,) * 3, m=m, sigma=sigma, ntransforms=C, kernel_evalmode=ev, backend=nufft.ROCBackend(0), spread_method="marching_ring")
        nufft.set_points(p, xs)
        CZ = torch.complex128 if T == torch.float64 else torch.complex64
        us = tuple(torch.empty(p.shape, dtype=CZ, device="cuda") for _ in range(C))
        nufft.exec_type1(us if C > 1 else us[0], p, vs if C > 1 else vs[0])
        info = p.info()
        out[f"u{ci}"] = torch.stack(us).cpu().numpy()
        out[f"h{ci}"] = np.array([info.ring_halo, info.ring_column[0], info.ring_column[1], {"lds_tiles": 1, "mfma_patches": 2, "marching_ring": 3}[p.spread_engine_used()]])
        # stage level: spread only, read the grid of the last component
        nufft.spread_from_points(p, vs if C > 1 else vs[0])
        out[f"g{ci}"] = nufft.oversampled_grid(p, C - 1).cpu().numpy()
    np.savez(sys.argv[2], **out)
    sys.exit(0)

res = {}
for name, env in (("h0", {"NUFFT_SMARCH_HALO": "0"}), ("h2", {"NUFFT_SMARCH_HALO": "2"}), ("h2u", {"NUFFT_SMARCH_HALO": "2", "NUFFT_SMARCH_HALO_FUSE": "0"})):
    f = f"/tmp/halo_{name}.npz"
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", f], env=e, capture_output=True, text=True)
    if r.returncode:
        print(name, "FAILED", r.stdout[-2000:], r.stderr[-3000:])
        sys.exit(1)
    res[name] = np.load(f)
for ci, c in enumerate(CASES):
    a = res["h0"][f"u{ci}"]
    line = [str(c), "halo/col/engine", res["h2"][f"h{ci}"].tolist()]
    for name in ("h2", "h2u"):
        b = res[name][f"u{ci}"]
        ga, gb = res["h0"][f"g{ci}"], res[name][f"g{ci}"]
        line.append(f"{name}: u {np.linalg.norm((a - b).ravel()) / np.linalg.norm(a.ravel()):.2e} grid {np.linalg.norm((ga - gb).ravel()) / np.linalg.norm(ga.ravel()):.2e}")
    print(*line, flush=True)
