"""GKS on the C5 shape (32 frames x 256^2, 15 angles per frame, space-time TV, lambda = 1e-2) on one GPU: iterations per second of
50-iteration solves."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trips_py_amd.operators import Radon2DParallel, BlockDiagOp, SpaceTimeDerivative
from trips_py_amd import solvers as S
N, nt, na = 256, int(sys.argv[1]) if len(sys.argv) > 1 else 32, 15
F = BlockDiagOp([Radon2DParallel(N, np.deg2rad(t * 3.0 + 12.0 * np.arange(na))) for t in range(nt)])
L = SpaceTimeDerivative(N, nt)
x = torch.rand(F.shape[1], device="cuda"); b = F.apply(x)
S.GKS(F, b, L, 3, 5, 1e-2, history=False); torch.cuda.synchronize()
r = []
for _ in range(5):
    t0 = time.perf_counter(); S.GKS(F, b, L, 3, 50, 1e-2, history=False); torch.cuda.synchronize(); r.append(50 / (time.perf_counter() - t0))
print("GKS it/s", " ".join("%.0f" % v for v in r))
