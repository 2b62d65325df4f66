"""Hybrid-LSQR on the 2-D Gaussian blur (512^2 and 2048^2, 9 x 9 PSF): fixed lambda / GCV, 60-iteration solves — the Golub-Kahan half
steps of an operator without hinted chains (trk_op_apply_axpby: the blur's own store carries the vector update and the norm)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trips_py_amd.operators import Blur2D
from trips_py_amd.solvers import Hybrid_LSQR
from trips_py_amd.problems import gauss_psf
for N in (512, 2048):
    psf, _ = gauss_psf((9, 9), (3, 3))
    A = Blur2D(psf, N, N)
    x = torch.rand(N * N, device="cuda"); b = A.apply(x)
    b = b + 0.01 * torch.randn_like(b) * b.norm() / b.numel() ** 0.5
    for reg in (1e-2, "gcv"):
        Hybrid_LSQR(A, b, 5, reg, x_true=x, history=False)
        ts = []
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            Hybrid_LSQR(A, b, 60, reg, x_true=x, history=False)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print(f"blur {N}^2 Hybrid_LSQR {reg}:", " ".join(f"{60/t:.0f}" for t in ts), "it/s")
