"""Does set_points get faster when the input points already arrive grouped by coarse bins?
(development probe for a two-level sort; not part of the product)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nufft_pkg import nufft
N, Np = 256, 10_000_000
plan = nufft.PlanNUFFT(torch.float64, (N, N, N), m=4, sigma=2.0, backend=nufft.ROCBackend(0), synchronise=True)
g = torch.Generator(device="cuda").manual_seed(1)
xs = [torch.rand(Np, dtype=torch.float64, device="cuda", generator=g) * (2 * np.pi) for _ in range(3)]
def timeit(pts, label):
    for _ in range(2): nufft.set_points(plan, pts)
    torch.cuda.synchronize(); ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); nufft.set_points(plan, pts); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    print(f"{label}: {np.median(ts):.3f} ms", flush=True)
timeit(tuple(xs), "random order")
for cl in (16, 64, 128):     # coarse bin edge in oversampled cells (512 per axis)
    cells = [torch.clamp((x / (2 * np.pi) * 512).long(), max=511) // cl for x in xs]
    nb = 512 // cl
    key = (cells[2] * nb + cells[1]) * nb + cells[0]
    perm = torch.argsort(key)
    pts = tuple(x[perm].contiguous() for x in xs)
    timeit(pts, f"grouped by {cl}^3-cell coarse bins ({nb**3} bins)")
