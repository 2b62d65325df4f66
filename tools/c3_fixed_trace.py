"""Hybrid-LSQR at fixed lambda on 512^2 x 180 (three solves of 50 iterations, x_true given): the program behind rocprofv3 --kernel-trace
for tools/trace_gaps.py.  argv[1] = 0: the LSQR update in its own launch (update_on_the_step=False)."""
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
b = b + 0.01 * torch.randn(b.numel(), device=dev, generator=g) * b.norm() / b.numel() ** 0.5
on = not (len(sys.argv) > 1 and sys.argv[1] == "0")
for rep in range(3):
    Hybrid_LSQR(A, b, 50, 1e-2, xt, update_on_the_step=on)
torch.cuda.synchronize()
