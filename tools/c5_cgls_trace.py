"""CGLS on the C5 shape (argv[1] frames, default 32) behind rocprofv3 --kernel-trace (tools/trace_gaps.py): argv[2] = 1 for the
one-all-reduce arrangement."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import numpy as np, torch
from trips_py_amd.operators import BlockDiagOp, Radon2DParallel
from trips_py_amd.solvers import CGLS
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 32
one = len(sys.argv) > 2 and sys.argv[2] == "1"
F = BlockDiagOp([Radon2DParallel(256, np.deg2rad(t + 12.0 * np.arange(15))) for t in range(nt)])
dev = F.engine.device
xt = torch.rand(F.shape[1], device=dev, generator=torch.Generator(device=dev).manual_seed(0))
b = F.apply(xt)
x0 = torch.zeros(F.shape[1], device=dev)
for _ in range(3):
    CGLS(F, b, x0, 100, 0, history=False, one_reduction=one)
torch.cuda.synchronize()
