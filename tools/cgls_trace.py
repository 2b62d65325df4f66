import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from trips_py_amd.operators import Blur2D
from trips_py_amd.problems import gauss_psf
from trips_py_amd.solvers import CGLSRun
N = 4096
A = Blur2D(gauss_psf((9, 9), (3, 3))[0], N, N)
x = torch.rand(N * N, device="cuda"); b = A.apply(x)
run = CGLSRun(A, b, torch.zeros_like(x), 220, None, False, defer_norms=True)
run.run(20); torch.cuda.synchronize(); run.run(200); torch.cuda.synchronize()
