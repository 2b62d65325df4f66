"""Where the fixed cost of one CGLS(...) call at 512^2 goes (bench.py's C2 figure times whole calls)."""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trips_py_amd.operators import Blur2D
from trips_py_amd.problems import gauss_psf
from trips_py_amd.solvers import CGLS
N = 512
A = Blur2D(gauss_psf((9, 9), (3, 3))[0], N, N)
x = torch.rand(N * N, device="cuda"); b = A.apply(x); x0 = torch.zeros_like(x)
for its in (100, 1):
    CGLS(A, b, x0, its, 0, x); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): CGLS(A, b, x0, its, 0, x)
    torch.cuda.synchronize()
    print(f"{its} iterations: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per call")
pr = cProfile.Profile(); pr.enable()
for _ in range(20): CGLS(A, b, x0, 100, 0, x)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(16)
