"""MMGKS + TV at 2048^2, one 30-iteration solve behind rocprofv3 (argv[1]: gcv | a number)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from trips_py_amd.operators import Blur2D, FirstDerivative2D
from trips_py_amd.problems import gauss_psf
from trips_py_amd import solvers as S
M = 2048
reg = "gcv" if (len(sys.argv) > 1 and sys.argv[1] == "gcv") else 1e-2
A = Blur2D(gauss_psf((9, 9), (3, 3))[0], M, M)
Ld = FirstDerivative2D(M, engine=A.engine)
xt = torch.rand(M * M, device="cuda"); bb = A.apply(xt)
bb = bb + 0.01 * torch.randn_like(bb) * bb.norm() / bb.numel() ** 0.5
for _ in range(2):
    S.MMGKS(A, bb, Ld, pnorm=2, qnorm=1, projection_dim=3, n_iter=30, regparam=reg, epsilon=0.1, history=False)
torch.cuda.synchronize()
