"""MMGKS + TV at 2048^2 / 4096^2 with regparam='gcv': the download of the 2 k^2 + 2 k Gram doubles per iteration through the mailbox (one
64-lane launch copying into pinned memory) against the tensor copy (DevScalars.HOST_BY_MAILBOX_MAX = 0)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trips_py_amd.operators import Blur2D, FirstDerivative2D
from trips_py_amd.problems import gauss_psf
from trips_py_amd import solvers as S
from trips_py_amd.engine import DevScalars
for M in (2048, 4096):
    A = Blur2D(gauss_psf((9, 9), (3, 3))[0], M, M)
    Ld = FirstDerivative2D(M, engine=A.engine)
    xt = torch.rand(M * M, device="cuda"); bb = A.apply(xt)
    bb = bb + 0.01 * torch.randn_like(bb) * bb.norm() / bb.numel() ** 0.5
    for cap in (4096, 256, 0):
        DevScalars.HOST_BY_MAILBOX_MAX = cap
        f = lambda: S.MMGKS(A, bb, Ld, pnorm=2, qnorm=1, projection_dim=3, n_iter=30, regparam="gcv", epsilon=0.1, history=False)
        f(); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print(f"MMGKS TV {M}^2 gcv, mailbox downloads up to {cap:5d} doubles: {30 / sorted(ts)[1]:8.1f} it/s")
    DevScalars.HOST_BY_MAILBOX_MAX = 4096
