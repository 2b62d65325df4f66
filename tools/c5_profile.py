import sys, os, cProfile, pstats, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trips_py_amd.operators import BlockDiagOp, Radon2DParallel, SpaceTimeDerivative
from trips_py_amd.solvers import CGLS, GKS
Nf, nt, na = 256, 4, 15
ops = [Radon2DParallel(Nf, np.deg2rad(t + 12.0 * np.arange(na))) for t in range(nt)]
F = BlockDiagOp(ops); L = SpaceTimeDerivative(Nf, nt)
eng = F.engine
xt = torch.rand(F.shape[1], device=eng.device); bl = F.apply(xt)
GKS(F, bl, L, 3, 3, 1e-2, history=False); torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter(); GKS(F, bl, L, 3, 50, 1e-2, history=False); torch.cuda.synchronize(); print("GKS it/s", 50 / (time.perf_counter() - t0))
pr = cProfile.Profile(); pr.enable(); GKS(F, bl, L, 3, 50, 1e-2, history=False); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
