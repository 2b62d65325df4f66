"""Fixed cost of one CGLS() call at 512^2 (tiled form): solves of 1, 10, 100, 400 iterations, with and without x_true; cProfile of one."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trips_py_amd.operators import Blur2D
from trips_py_amd.problems import gauss_psf
from trips_py_amd.solvers import CGLS
N = 512
A = Blur2D(gauss_psf((9, 9), (3, 3))[0], N, N)
dev = A.engine.device
xt = torch.rand(N * N, device=dev); b = A.apply(xt); x0 = torch.zeros(N * N, device=dev)
for kw in ({"x_true": xt}, {}):
    for its in (1, 10, 100, 400):
        CGLS(A, b, x0, its, 0, history=False, **kw); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            CGLS(A, b, x0, its, 0, history=False, **kw)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print(f"x_true={'x_true' in kw!s:5s} its={its:4d}: {dt*1e3:7.3f} ms per solve = {dt/its*1e6:7.2f} us per iteration")
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    CGLS(A, b, x0, 100, 0, history=False, x_true=xt)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
