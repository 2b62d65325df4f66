"""Host time of a Hybrid_LSQR iteration with an automatic lambda, by section (perf_counter around the calls the loop makes)."""
import sys, os, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trips_py_amd.operators import Radon2DParallel
from trips_py_amd.solvers import Hybrid_LSQR
HL = sys.modules["trips_py_amd.solvers.Hybrid_LSQR"]
from trips_py_amd import krylov
N = 512
R = Radon2DParallel(N, np.linspace(0, np.pi, 180, endpoint=False))
x = torch.rand(N * N, device="cuda"); b = R.apply(x)
b = b + 0.01 * torch.randn_like(b) * b.norm() / b.numel() ** 0.5
reg = sys.argv[1] if len(sys.argv) > 1 else "gcv"
kw = {"delta": float(0.01 * b.norm())} if reg == "dp" else {}
acc = collections.defaultdict(float)
def wrap(obj, name, tag):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[tag] += time.perf_counter() - t
    setattr(obj, name, g)
wrap(krylov.GKState, "absorb", "absorb (wait for the norms)")
wrap(krylov.GKState, "step_prefetch", "step_prefetch (enqueue a step + download)")
wrap(HL._Searcher, "collect", "collect lambda from the worker")
wrap(HL._Searcher, "post_gcv", "post gcv search")
wrap(HL._Searcher, "post_dp", "post dp search")
eng = R.engine
wrap(type(eng), "bidiag_tikhonov", "bidiag_tikhonov")
wrap(type(eng), "gemv_n_err", "gemv_n_err")
Hybrid_LSQR(R, b, 100, reg, x_true=x, history=False, **kw)
torch.cuda.synchronize(); acc.clear()
t0 = time.perf_counter()
Hybrid_LSQR(R, b, 100, reg, x_true=x, history=False, **kw)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{reg}: loop returned after {(t1-t0)*1e3:.2f} ms, device done after {(t2-t0)*1e3:.2f} ms")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:45s} {v*1e4:7.1f} us / iteration")
print(f"  {'everything else in the loop':45s} {((t1-t0)-sum(acc.values()))*1e4:7.1f} us / iteration")
