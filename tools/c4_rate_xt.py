"""tools/c4_rate.py with x_true given (relError per iteration: the error norm rides the pass that forms x = V y)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trips_py_amd.operators import Blur2D, FirstDerivative2D  # noqa: E402
from trips_py_amd.problems import gauss_psf  # noqa: E402
from trips_py_amd.solvers import MMGKS  # noqa: E402
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
A = Blur2D(gauss_psf((9, 9), (3, 3))[0], N, N)
L = FirstDerivative2D(N)
dev = A.engine.device
x = torch.rand(N * N, device=dev, generator=torch.Generator(device=dev).manual_seed(0))
b = A.apply(x)
b = b + 0.01 * torch.randn(N * N, device=dev, generator=torch.Generator(device=dev).manual_seed(1)) * b.norm() / N
for kw in ({}, {"fused_error_norm": False}):
    MMGKS(A, b, L, 2, 1, 3, 4, 1e-2, x, history=False, **kw)
    torch.cuda.synchronize()
    rates = []
    for _ in range(3):
        t0 = time.perf_counter()
        xx, info = MMGKS(A, b, L, 2, 1, 3, 30, 1e-2, x, history=False, **kw)
        torch.cuda.synchronize()
        rates.append(30 / (time.perf_counter() - t0))
    print(kw, "MMGKS(x_true) it/s:", [round(r, 1) for r in rates], "relError[-1] %.9f" % info["relError"][-1])
