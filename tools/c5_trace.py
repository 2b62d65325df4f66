import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trips_py_amd.operators import BlockDiagOp, Radon2DParallel, SpaceTimeDerivative
from trips_py_amd.solvers import CGLS, GKS
Nf, nt, na = 256, 4, 15
F = BlockDiagOp([Radon2DParallel(Nf, np.deg2rad(t + 12.0 * np.arange(na))) for t in range(nt)]); L = SpaceTimeDerivative(Nf, nt)
xt = torch.rand(F.shape[1], device="cuda"); bl = F.apply(xt)
which = sys.argv[1] if len(sys.argv) > 1 else "gks"
if which == "gks":
    GKS(F, bl, L, 3, 3, 1e-2, history=False); torch.cuda.synchronize()
    GKS(F, bl, L, 3, 50, 1e-2, history=False); torch.cuda.synchronize()
else:
    CGLS(F, bl, torch.zeros_like(xt), 5, 0, history=False); torch.cuda.synchronize()
    CGLS(F, bl, torch.zeros_like(xt), 100, 0, history=False); torch.cuda.synchronize()
