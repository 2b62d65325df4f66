import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trips_py_amd.operators import Radon2DParallel
from trips_py_amd.solvers import Hybrid_LSQR
N = 512
R = Radon2DParallel(N, np.linspace(0, np.pi, 180, endpoint=False))
x = torch.rand(N * N, device="cuda"); b = R.apply(x)
Hybrid_LSQR(R, b, 5, 1e-2, history=False); torch.cuda.synchronize()
Hybrid_LSQR(R, b, 100, 1e-2, history=False); torch.cuda.synchronize()
