import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import numpy as np, torch
from trips_py_amd.operators import Radon2DParallel
from trips_py_amd.solvers import Hybrid_LSQR
N = 512
A = Radon2DParallel(N, np.linspace(0, np.pi, 180, endpoint=False))
dev = A.engine.device
g = torch.Generator(device=dev).manual_seed(1)
xt = torch.rand(N * N, device=dev, generator=g)
b = A.apply(xt)
e = torch.randn(b.numel(), device=dev, generator=g)
delta = 0.01 * float(b.norm())
b = b + e * (delta / e.norm())
reg = sys.argv[1] if len(sys.argv) > 1 else "dp"
kw = {"delta": delta} if reg == "dp" else {}
for rep in range(3):
    Hybrid_LSQR(A, b, 50, reg, xt, **kw)
torch.cuda.synchronize()
