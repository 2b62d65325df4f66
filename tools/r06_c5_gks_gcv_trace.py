import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trips_py_amd.operators import Radon2DParallel, BlockDiagOp, SpaceTimeDerivative
from trips_py_amd import solvers as S
N, nt, na = 256, 32, 15
F = BlockDiagOp([Radon2DParallel(N, np.deg2rad(t * 3.0 + 12.0 * np.arange(na))) for t in range(nt)])
L = SpaceTimeDerivative(N, nt)
x = torch.rand(F.shape[1], device="cuda"); b = F.apply(x)
S.GKS(F, b, L, 3, 5, "gcv", history=False); torch.cuda.synchronize()
S.GKS(F, b, L, 3, 50, "gcv", history=False); torch.cuda.synchronize()
