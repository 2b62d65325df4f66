"""Micro-driver for profiling the blur kernel alone: N x N image, 9x9 Gaussian PSF, R forward applies
(with and without the fused sum of squares).  Used under rocprofv3 (kernel trace / PMC passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from trips_py_amd.operators import Blur2D
from trips_py_amd.problems import gauss_psf

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
R = int(sys.argv[2]) if len(sys.argv) > 2 else 20
K = int(sys.argv[3]) if len(sys.argv) > 3 else 9
psf, _ = gauss_psf((K, K), (K / 3.0, K / 3.0))
A = Blur2D(psf, N, N)
eng = A.engine
x = torch.randn(N * N, device=eng.device)
y = torch.empty_like(x)
S = eng.scalars(1)
for _ in range(R):
    A.apply(x, out=y)
for _ in range(R):
    A.apply(x, out=y, sumsq=S[0:1])
# reference points: a plain device copy and an axpby of the same size
for _ in range(R):
    y.copy_(x)
for _ in range(R):
    eng.axpby(1.0, x, 2.0, y, y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for name, fn in [("blur", lambda: A.apply(x, out=y)), ("blur+sumsq", lambda: A.apply(x, out=y, sumsq=S[0:1])),
                 ("copy", lambda: y.copy_(x)), ("axpby", lambda: eng.axpby(1.0, x, 2.0, y, y))]:
    e0.record()
    for _ in range(R):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:12s} {e0.elapsed_time(e1) / R * 1e3:8.2f} us/launch (back-to-back, N={N})")
