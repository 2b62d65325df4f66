"""How far do the two arrangements of CGLS drift from a float64 run of the same recurrence?  (a) the recurrence as written — two
reductions per iteration, w = A p formed fresh; (b) the one-all-reduce arrangement of csrc/cgls_sharded.hip (w by recurrence).
Problem: the dynamic parallel-beam shape of C5 in miniature (4 frames x 64^2, 15 angles per frame: under-determined, 1 % noise), and
the C5 shape itself; float64 reference = the oracle's CGLS over the oracle's projector (CPU).  GPU box."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cpu_ref as O  # noqa: E402
from trips_py_amd import solvers as S  # noqa: E402
from trips_py_amd.operators import BlockDiagOp, Radon2DParallel  # noqa: E402


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64).ravel(), np.asarray(b, dtype=np.float64).ravel()
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


for N, nt, its in ((64, 4, 40), (256, 32, 60)):
    angs = [np.deg2rad(t + 12.0 * np.arange(15)) for t in range(nt)]
    F = BlockDiagOp([Radon2DParallel(N, a) for a in angs])
    Fo = O.BlockDiag([O.Radon2D(N, a) for a in angs])
    rng = np.random.default_rng(0)
    xt = rng.random(F.shape[1])
    b = Fo @ xt
    e = rng.standard_normal(b.size)
    b = (b + 0.01 * np.linalg.norm(b) / np.linalg.norm(e) * e).astype(np.float32).astype(np.float64)
    x0 = np.zeros(F.shape[1])
    xo, io = O.cgls(Fo, b.reshape(-1, 1), x0.reshape(-1, 1), its, 0, xt.reshape(-1, 1))
    xa, ia = S.CGLS(F, b, x0, its, 0, xt, one_reduction=False)
    xb, ib = S.CGLS(F, b, x0, its, 0, xt, one_reduction=True)
    print(f"{nt} x {N}^2, {its} iterations: relError of the float64 run {io['relError'][0]:.3f} -> {min(io['relError']):.3f} (min) -> {io['relError'][-1]:.3f}")
    for name, info in (("two reductions", ia), ("one reduction", ib)):
        d = [rel(h, ho) for h, ho in zip(info["xHistory"], io["xHistory"])]
        print(f"  {name:15s} vs float64, iterate 1/5/10/20/30/{its}:", " ".join(f"{d[k - 1]:.1e}" for k in (1, 5, 10, 20, 30, its) if k <= its))
    d = [rel(h, ho) for h, ho in zip(ib["xHistory"], ia["xHistory"])]
    print("  one vs two reductions:               ", " ".join(f"{d[k - 1]:.1e}" for k in (1, 5, 10, 20, 30, its) if k <= its))
