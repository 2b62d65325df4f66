"""gcv Hybrid-GMRES (one library call per iteration) on the 512^2 blur behind rocprofv3 --kernel-trace (tools/trace_gaps.py lists the last launches)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from trips_py_amd.operators import Blur2D
from trips_py_amd.problems import gauss_psf
from trips_py_amd.solvers import Hybrid_GMRES
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
A = Blur2D(gauss_psf((9, 9), (3, 3))[0], N, N)
dev = A.engine.device
x = torch.rand(N * N, device=dev); b = A.apply(x)
Hybrid_GMRES(A, b, 20, "gcv", x); torch.cuda.synchronize()
Hybrid_GMRES(A, b, 60, "gcv", x); torch.cuda.synchronize()
