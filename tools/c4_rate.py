"""C4 alone (MMGKS + TV, 4096^2, 30 iterations, lambda = 1e-2) as the bench times it: iterations per second, three timed solves.
For A/B runs over the library's environment knobs and for `rocprofv3 --kernel-trace --stats -- python3 tools/c4_rate.py`."""
import os
import sys
import time

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
MMGKS(A, b, L, 2, 1, 3, 4, 1e-2, history=False)
torch.cuda.synchronize()
rates = []
for _ in range(3):
    t0 = time.perf_counter()
    MMGKS(A, b, L, 2, 1, 3, 30, 1e-2, history=False)
    torch.cuda.synchronize()
    rates.append(30 / (time.perf_counter() - t0))
print("knobs", {k: v for k, v in os.environ.items() if k.startswith("TRK_")}, "MMGKS it/s:", [round(r, 1) for r in rates])
