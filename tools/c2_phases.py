"""Phases of one CGLS() call at 512^2 with x_true (tiled form): constructor, enqueue of 100 iterations, wait + download, info."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trips_py_amd.operators import Blur2D
from trips_py_amd.problems import gauss_psf
C = sys.modules.get("trips_py_amd.solvers.CGLS") or __import__("trips_py_amd.solvers.CGLS", fromlist=["x"])
C = sys.modules["trips_py_amd.solvers.CGLS"]
N = 512
A = Blur2D(gauss_psf((9, 9), (3, 3))[0], N, N)
dev = A.engine.device
xt = torch.rand(N * N, device=dev); b = A.apply(xt); x0 = torch.zeros(N * N, device=dev)
acc = collections.defaultdict(float)
def wrap(cls, name, tag):
    f = getattr(cls, name)
    def g(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[tag] += time.perf_counter() - t
    setattr(cls, name, g)
wrap(C.CGLSRunFused, "__init__", "constructor (buffers, t0 = A^T(b - A x0), norms)")
wrap(C.CGLSRunFused, "run", "enqueue 100 iterations (one C call)")
wrap(C.CGLSRunFused, "rows", "finish: two reduction launches, wait for the device, download")
for _ in range(3):
    C.CGLS(A, b, x0, 100, 0, x_true=xt, history=False)
torch.cuda.synchronize(); acc.clear()
t0 = time.perf_counter()
for _ in range(20):
    C.CGLS(A, b, x0, 100, 0, x_true=xt, history=False)
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) / 20
print(f"per solve {tot*1e3:.3f} ms")
for k, v in acc.items():
    print(f"  {k:70s} {v/20*1e6:8.1f} us")
print(f"  {'the rest of CGLS() (formatting, info lists)':70s} {(tot - sum(acc.values())/20)*1e6:8.1f} us")
