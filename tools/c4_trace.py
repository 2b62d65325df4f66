import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trips_py_amd.operators import Blur2D, FirstDerivative2D
from trips_py_amd.problems import gauss_psf
from trips_py_amd.solvers import MMGKS
N = 4096
A = Blur2D(gauss_psf((9, 9), (3, 3))[0], N, N)
L = FirstDerivative2D(N)
x = torch.rand(N * N, device="cuda"); b = A.apply(x)
MMGKS(A, b, L, 2, 1, 3, 4, 1e-2, history=False); torch.cuda.synchronize()
MMGKS(A, b, L, 2, 1, 3, 30, 1e-2, history=False); torch.cuda.synchronize()
