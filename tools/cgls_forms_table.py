"""Per-iterate distances of the two CGLS arrangements from the float64 oracle and from each other, on the three problems of
tests/test_gpu_dist.py::test_one_reduction_cgls_equals_the_recurrence_as_written (GPU box)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import relerr  # noqa: E402
from oracle import cpu_ref as O  # noqa: E402
from trips_py_amd import solvers as S  # noqa: E402
from trips_py_amd.operators import Blur2D, BlockDiagOp, Radon2DParallel  # noqa: E402
from trips_py_amd.problems import gauss_psf  # noqa: E402

for case in ("tomo_dynamic", "blur96", "tomo_static"):
    if case == "tomo_dynamic":
        angs = [np.deg2rad(t + 12.0 * np.arange(15)) for t in range(4)]
        A, Ao = BlockDiagOp([Radon2DParallel(64, a) for a in angs]), O.BlockDiag([O.Radon2D(64, a) for a in angs])
    elif case == "blur96":
        psf = gauss_psf((9, 9), (3, 3))[0]
        A, Ao = Blur2D(psf, 96, 96), O.Blur2D(psf, 96, 96)
    else:
        ang = np.linspace(0, np.pi, 45, endpoint=False)
        A, Ao = Radon2DParallel(128, ang), O.Radon2D(128, ang)
    rng = np.random.default_rng(1)
    xt = rng.random(A.shape[1])
    b = Ao @ xt
    e = rng.standard_normal(b.size)
    b = (b + 0.01 * np.linalg.norm(b) / np.linalg.norm(e) * e).astype(np.float32).astype(np.float64)
    x0 = np.zeros(A.shape[1])
    its = 40
    xo, io = O.cgls(Ao, b.reshape(-1, 1), x0.reshape(-1, 1), its, 0, xt.reshape(-1, 1))
    xa, ia = S.CGLS(A, b, x0, its, 0, xt, one_reduction=False, tiled=False, fused=False)
    xb, ib = S.CGLS(A, b, x0, its, 0, xt, one_reduction=True)
    print(case, "k: two_vs_64 one_vs_64 one_vs_two")
    for k in range(its):
        ra, rb, ro = ia["xHistory"][k], ib["xHistory"][k], io["xHistory"][k]
        print(f"  {k + 1:2d}: {relerr(ra, ro):.1e} {relerr(rb, ro):.1e} {relerr(rb, ra):.1e}")
