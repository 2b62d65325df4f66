"""CGLS on the C5 shape (frames x 256^2, 15 angles per frame) on one GPU, both arrangements (the recurrence as written / one
all-reduce per iteration), for `frames` = 32 (one rank's problem at N = 1) ... 4 (one rank's share at N = 8): iterations per second.
usage: python3 tools/c5_cgls_rate.py [frames ...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trips_py_amd.operators import BlockDiagOp, Radon2DParallel  # noqa: E402
from trips_py_amd.solvers import CGLS  # noqa: E402

for nt in ([int(a) for a in sys.argv[1:]] or [32, 16, 8, 4]):
    F = BlockDiagOp([Radon2DParallel(256, np.deg2rad(t + 12.0 * np.arange(15))) for t in range(nt)])
    dev = F.engine.device
    xt = torch.rand(F.shape[1], device=dev, generator=torch.Generator(device=dev).manual_seed(0))
    b = F.apply(xt)
    x0 = torch.zeros(F.shape[1], device=dev)
    out = []
    for one in (False, True):
        CGLS(F, b, x0, 100, 0, history=False, one_reduction=one)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            CGLS(F, b, x0, 100, 0, history=False, one_reduction=one)
        torch.cuda.synchronize()
        out.append(500 / (time.perf_counter() - t0))
    print(f"{nt:3d} frames: two reductions {out[0]:8.0f} it/s, one reduction {out[1]:8.0f} it/s")
