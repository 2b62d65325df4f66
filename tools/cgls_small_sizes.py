"""CGLS(100 iterations, tol = 0, history off) on small blur problems: tiled two-launch form vs the streaming forms."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from trips_py_amd.operators import Blur2D
from trips_py_amd.problems import gauss_psf
from trips_py_amd.solvers import CGLS
psf = gauss_psf((9, 9), (3, 3))[0]
for N in [int(a) for a in sys.argv[1:]] or [128, 256, 512, 768, 1024]:
    A = Blur2D(psf, N, N)
    dev = A.engine.device
    xt = torch.rand(N * N, device=dev)
    b = A.apply(xt)
    x0 = torch.zeros(N * N, device=dev)
    out = []
    for kw in ({"tiled": True}, {"tiled": False}):
        CGLS(A, b, x0, 100, 0, history=False, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            CGLS(A, b, x0, 100, 0, history=False, **kw)
        torch.cuda.synchronize()
        out.append(500 / (time.perf_counter() - t0))
    print(f"N={N}: tiled {out[0]:9.0f} it/s   streaming {out[1]:9.0f} it/s")
