"""Hybrid-GMRES with gcv on the 512^2 blur: iterations/s of 60-iteration solves by the number of worker threads that run the projected
problems side by side (search_workers), and of the Python loop (c_loop=False)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trips_py_amd.operators import Blur2D
from trips_py_amd.problems import gauss_psf
from trips_py_amd.solvers import Hybrid_GMRES
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
A = Blur2D(gauss_psf((9, 9), (3, 3))[0], N, N)
dev = A.engine.device
x = torch.rand(N * N, device=dev, generator=torch.Generator(device=dev).manual_seed(0))
b = A.apply(x)
e = torch.randn(N * N, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
b = b + e * (0.01 * float(b.norm()) / e.norm())
ref = None
for name, kw in [("python loop", {"c_loop": False})] + [(f"one call per iteration, {w} worker(s)", {"search_workers": w}) for w in (1, 2, 3, 4, 6)]:
    Hybrid_GMRES(A, b, 20, "gcv", x, **kw)
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        xx, info = Hybrid_GMRES(A, b, 60, "gcv", x, **kw)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ph = []
    if "c_loop" not in kw:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        Hybrid_GMRES(A, b, 60, "gcv", x, host_phases=ph, **kw)
        torch.cuda.synchronize(); tt = time.perf_counter() - t0
        ph = [f"{1e6 * v / 60:.1f}" for v in ph] + [f"of {1e6 * tt / 60:.1f}"]
    lam = np.array(info["regParam_history"])
    if ref is None:
        ref = lam
    print(f"{name:40s}", " ".join(f"{60 / t:6.0f}" for t in ts), "it/s   max |lambda - python loop's| / lambda %.1e" % float(np.max(np.abs(lam - ref)[1:] / ref[1:])),
          ("  us per iteration: wait step | enqueue | wait worker | post | launch x: " + " ".join(ph)) if ph else "")
