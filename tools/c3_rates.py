import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from trips_py_amd.operators import Radon2DParallel
from trips_py_amd.solvers import Hybrid_LSQR
N = 512
A = Radon2DParallel(N, np.linspace(0, np.pi, 180, endpoint=False))
dev = A.engine.device
g = torch.Generator(device=dev).manual_seed(1)
xt = torch.rand(N * N, device=dev, generator=g)
b = A.apply(xt)
e = torch.randn(b.numel(), device=dev, generator=g)
delta = 0.01 * float(b.norm())
b = b + e * (delta / e.norm())
for reg, kw in (("dp", {"delta": delta}), ("gcv", {}), (1e-2, {})):
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        x, info = Hybrid_LSQR(A, b, 50, reg, xt, **kw)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(reg, "%.0f it/s" % (50 / dt), info["relError"][-1], info["regParam"])
if len(sys.argv) > 1 and sys.argv[1] == "profile":
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    Hybrid_LSQR(A, b, 50, "dp", xt, delta=delta)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)
