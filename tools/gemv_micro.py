"""The basis kernels of the projection solvers alone, at the C4 shape (n = 4096^2 fp32 rows): out = base - V c (trk_gemv_n, + sum of
squares), x = V y (trk_gemv_n), V^T [r, r2] (trk_gemv_t2), V^T [4 right-hand sides] (trk_gemv_tn), V^T r (trk_gemv_t) — time per
launch and bytes of basis streamed per second, over basis sizes k.  Knobs (environment, read by the library):
TRK_GEMVN_UNROLL = 4 / 8 / 16 (basis rows requested together), TRK_GEMVN_GRID (blocks per CU), TRK_GEMVT_PER_CU (blocks per CU of the
transposed family), TRK_NT (cache hints).   usage: [GEMV_MICRO_K=4,8,...] python3 tools/gemv_micro.py [N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trips_py_amd.engine import HipEngine  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = N * N
torch.cuda.set_device(0)
eng = HipEngine()
KS = tuple(int(v) for v in os.environ.get("GEMV_MICRO_K", "4,8,12,18,24,32").split(","))     # basis sizes (e.g. GEMV_MICRO_K=16,32,52 at C5's n)
KMAX = max(KS) + 1
V = eng.empty_basis(KMAX, n)
V.normal_()
r, r2, r3, r4, out = (torch.randn(n, device="cuda") for _ in range(5))
y = eng.scalars(KMAX)
y.set(0, [0.01 * (j + 1) for j in range(KMAX)])
h = eng.scalars(4 * KMAX)
ss = eng.scalars(1)


def timeit(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3          # us


print("knobs:", {k: v for k, v in os.environ.items() if k.startswith("TRK_")})
print(f"{'k':>3} | {'gemv_n base+ss':>16} | {'gemv_n plain':>16} | {'gemv_t2':>16} | {'gemv_tn(4)':>16} | {'gemv_t':>16}   (us, TB/s of 4 k n + vectors)")
for k in KS:
    cells = []
    for name, fn, nbytes in (
            ("n_base", lambda: eng.gemv_n(V, k, y.ref(0), out, a=1.0, base=r, s=-1.0, sumsq=ss.ref(0)), 4.0 * n * (k + 2)),
            ("n", lambda: eng.gemv_n(V, k, y.ref(0), out), 4.0 * n * (k + 1)),
            ("t2", lambda: eng.gemv_t2(V, k, r, r2, h.ref(0)), 4.0 * n * (k + 2)),
            ("tn4", lambda: eng.gemv_tn(V, k, [r, r2, r3, r4], h.ref(0)), 4.0 * n * (k + 4)),
            ("t", lambda: eng.gemv_t(V, k, r, h.ref(0)), 4.0 * n * (k + 1))):
        us = timeit(fn)
        cells.append(f"{us:8.1f} {nbytes / us / 1e6:6.2f}")
    print(f"{k:3d} | " + " | ".join(f"{c:>16}" for c in cells))
