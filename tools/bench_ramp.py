"""How long does the headline loop take to reach its steady state?  The bench's problem (4096^2 CGLS, the form CGLS() picks),
from a cold start: forward-blur kernel duration (dispatch events) and wall time per iteration over consecutive windows.
usage: python3 tools/bench_ramp.py [total_iters] [idle_ms_before_start]"""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402

total = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
idle_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
torch.cuda.set_device(0)
from trips_py_amd.operators import Blur2D  # noqa: E402
from trips_py_amd.problems import gauss_psf  # noqa: E402
from trips_py_amd.solvers import CGLSRun  # noqa: E402

N = 4096
n = N * N
psf, _ = gauss_psf((9, 9), (3, 3))
A = Blur2D(psf, N, N)
eng = A.engine
xt = torch.rand(n, device=eng.device)
b = A.apply(xt)
x0 = torch.zeros(n, device=eng.device)
run = CGLSRun(A, b, x0, total, x_true=None, history=False, defer_norms=True)
torch.cuda.synchronize()
if idle_ms:
    time.sleep(idle_ms * 1e-3)
t = bench.KernelTimer(A, total + 4, 0)
t.attach()
edges = [0, 5, 25, 50, 100, 200, 400, 800, 1600, 3200, 6400, 12800]
edges = [e for e in edges if e < total] + [total]
walls = []
for lo, hi in zip(edges[:-1], edges[1:]):
    t0 = time.perf_counter()
    run.run(hi - lo)
    torch.cuda.synchronize()
    walls.append((time.perf_counter() - t0) / (hi - lo) * 1e6)
t.detach()
us = t.read() * 1e3
print(f"idle before start {idle_ms} ms; {len(us)} forward launches timed")
for (lo, hi), w in zip(zip(edges[:-1], edges[1:]), walls):
    s = us[lo:hi]
    print(f"iters {lo:5d}-{hi:5d}: fwd kernel avg {s.mean():6.2f} med {np.median(s):6.2f} p90 {np.percentile(s, 90):6.2f} "
          f"min {s.min():6.2f} max {s.max():6.2f} us | wall {w:7.1f} us/iter ({1e6 / w:7.0f} it/s)")
