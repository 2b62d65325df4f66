"""How fast does the host enqueue a CGLS stretch (trk_cgls_iterate_fused) compared with how fast the GPU retires it?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from trips_py_amd.operators import Blur2D
from trips_py_amd.problems import gauss_psf
from trips_py_amd.solvers import CGLSRunFused, CGLSRun
for N in (256, 512, 1024):
    A = Blur2D(gauss_psf((9, 9), (3, 3))[0], N, N)
    x = torch.rand(N * N, device="cuda"); b = A.apply(x)
    for cls, kw in ((CGLSRunFused, {}), (CGLSRun, {"defer_norms": True})):
        run = cls(A, b, torch.zeros_like(x), 1100, None, False, **kw)
        run.run(100); torch.cuda.synchronize()
        t0 = time.perf_counter(); run.run(1000); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"N={N} {cls.__name__:13s}: host enqueue {1e6*(t1-t0)/1000:6.2f} us/iter, GPU retire {1e6*(t2-t0)/1000:6.2f} us/iter")
