"""C4 (MMGKS + TV, 4096^2, 30 iterations — the solve bench.py times) against the float64 restatement of MMGKS.py:37-128 that
tests/test_gpu_mmgks_fullsize.py pins to the oracle at 256^2: relative distance of every iterate.  GPU box; ~20 GB of device memory."""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import test_gpu_mmgks_fullsize as T  # noqa: E402
from test_gpu_fullsize import make_problem  # noqa: E402
from trips_py_amd import solvers as S  # noqa: E402
from trips_py_amd.operators import Blur2D, FirstDerivative2D  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n_iter = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda")
psf, xt, b = make_problem(N, dev)
b32 = b.float()
t0 = time.time()
hist = T.mmgks64(psf, N, b32.double(), 3, n_iter, 1e-2, 0.1, 2, 1)
torch.cuda.synchronize()
t1 = time.time()
x, info = S.MMGKS(Blur2D(psf, N, N), b32, FirstDerivative2D(N), 2, 1, 3, n_iter, 1e-2, epsilon=0.1)
torch.cuda.synchronize()
t2 = time.time()
errs = [float(torch.linalg.norm(info["xHistory"][k].reshape(-1).double() - hist[k]) / torch.linalg.norm(hist[k])) for k in range(len(hist))]
print(f"C4 MMGKS {N}^2, {n_iter} iterations (basis 3 -> {3 + n_iter}): float64 checker {t1 - t0:.1f} s, engine {t2 - t1:.2f} s, "
      f"peak device memory {torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GiB")
print("per iterate:", " ".join(f"{e:.1e}" for e in errs))
print(f"max {max(errs):.2e} at iterate {1 + errs.index(max(errs))}")
