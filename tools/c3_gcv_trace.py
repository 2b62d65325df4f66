import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trips_py_amd.operators import Radon2DParallel
from trips_py_amd.solvers import Hybrid_LSQR
N = 512
R = Radon2DParallel(N, np.linspace(0, np.pi, 180, endpoint=False))
x = torch.rand(N * N, device="cuda"); b = R.apply(x)
b = b + 0.01 * torch.randn_like(b) * b.norm() / b.numel() ** 0.5
reg = sys.argv[1] if len(sys.argv) > 1 else "gcv"
kw = {"delta": float(0.01 * b.norm())} if reg == "dp" else {}
Hybrid_LSQR(R, b, 100, reg, x_true=x, history=False, **kw); torch.cuda.synchronize()
Hybrid_LSQR(R, b, 100, reg, x_true=x, history=False, **kw); torch.cuda.synchronize()
