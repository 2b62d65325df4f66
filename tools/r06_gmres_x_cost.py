"""What forming every iterate costs Hybrid-GMRES on the 512^2 blur: 60-iteration solves with x_true (relError every iteration: x = V y every
iteration, one launch over k rows) against history=False without x_true (x formed once, after the loop).  Upper bound of what riding x = V y
on the next step's orthogonalisation pass could give."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trips_py_amd.operators import Blur2D
from trips_py_amd.problems import gauss_psf
from trips_py_amd import solvers as S
N = 512
A = Blur2D(gauss_psf((9, 9), (3, 3))[0], N, N)
dev = A.engine.device
xt = torch.rand(N * N, device=dev, generator=torch.Generator(device=dev).manual_seed(0))
b = A.apply(xt)
b = b + 0.01 * torch.randn(N * N, device=dev, generator=torch.Generator(device=dev).manual_seed(1)) * b.norm() / N
def rate(fn, its=60, reps=7):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return its / sorted(ts)[len(ts) // 2]
for reg in (1e-2, "gcv"):
    a = rate(lambda: S.Hybrid_GMRES(A, b, 60, reg, xt))
    c = rate(lambda: S.Hybrid_GMRES(A, b, 60, reg, history=False))
    print(f"Hybrid_GMRES 512^2 blur, regparam={reg!r}: every iterate formed {a:8.0f} it/s   only the last {c:8.0f} it/s")
