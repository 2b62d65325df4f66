"""cProfile of one Hybrid-LSQR / GKS / MMGKS solve on the GPU box: where does the HOST time go?"""
import sys, os, cProfile, pstats, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trips_py_amd.operators import Radon2DParallel, Blur2D, FirstDerivative2D
from trips_py_amd.problems import gauss_psf
from trips_py_amd.solvers import Hybrid_LSQR, GKS, MMGKS, Hybrid_GMRES

which = sys.argv[1] if len(sys.argv) > 1 else "lsqr"
reg = sys.argv[2] if len(sys.argv) > 2 else "1e-2"
reg = float(reg) if reg[0].isdigit() else reg
N = int(sys.argv[3]) if len(sys.argv) > 3 else 512
its = int(sys.argv[4]) if len(sys.argv) > 4 else 100
if which in ("lsqr",):
    A = Radon2DParallel(N, np.linspace(0, np.pi, 180, endpoint=False))
else:
    A = Blur2D(gauss_psf((9, 9), (3, 3))[0], N, N)
eng = A.engine
x = torch.rand(N * N, device=eng.device)
b = A.apply(x)
b = b + 0.01 * torch.linalg.norm(b) / np.sqrt(b.numel()) * torch.randn_like(b)
L = FirstDerivative2D(N)
kw = {"delta": float(0.01 * torch.linalg.norm(b))} if reg == "dp" else {}
def run(n):
    if which in ("lsqr", "lsqrb"): return Hybrid_LSQR(A, b, n, reg, history=False, **kw)
    if which == "gmres": return Hybrid_GMRES(A, b, n, reg, history=False, **kw)
    if which == "gks": return GKS(A, b, L, 3, n, reg, history=False, **kw)
    if which == "mmgks": return MMGKS(A, b, L, 2, 1, 3, n, reg, history=False, **kw)
if os.environ.get("GK_NORMALIZED"):      # timing comparison only (y is then scaled wrongly): the older normalised storage
    H = sys.modules["trips_py_amd.solvers.Hybrid_LSQR"]
    from trips_py_amd.krylov import GKState
    H.GKState = lambda A, b, n, normalized=False: GKState(A, b, n)
run(5); torch.cuda.synchronize()
t0 = time.perf_counter(); run(its); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"{which} reg={reg} N={N}: {its} iterations in {dt*1e3:.1f} ms = {its/dt:.0f} it/s")
pr = cProfile.Profile(); pr.enable(); run(its); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
