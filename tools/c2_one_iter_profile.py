"""Where the fixed cost of a CGLS() call at 512^2 goes: cProfile of 200 one-iteration calls (torch in, x_true given)."""
import sys, os, cProfile, pstats
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
for _ in range(20):
    CGLS(A, b, x0, 1, 0, xt)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    CGLS(A, b, x0, 1, 0, xt)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
