"""CGLS on the C2 shape (512^2 blur, 100 iterations, x_true) behind rocprofv3 --kernel-trace (tools/trace_gaps.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from trips_py_amd.operators import Blur2D
from trips_py_amd.problems import gauss_psf
from trips_py_amd.solvers import CGLS
N = 512
A = Blur2D(gauss_psf((9, 9), (3, 3))[0], N, N)
dev = A.engine.device
xt = torch.rand(N * N, device=dev)
b = A.apply(xt)
x0 = torch.zeros(N * N, device=dev)
for _ in range(3):
    CGLS(A, b, x0, 100, 0, xt)
torch.cuda.synchronize()
