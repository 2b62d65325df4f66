"""GKS with ONE pass over the basis for the new vector and the next iterate (trk_gemv_orth_iterate, late round 6) against a pass each
(fused_orth_iterate=False): iterations/s on C5's shape and on 2048^2 / 4096^2 blur + TV, fixed lambda and 'gcv', and the distance of the results."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trips_py_amd.operators import Radon2DParallel, BlockDiagOp, SpaceTimeDerivative, Blur2D, FirstDerivative2D
from trips_py_amd.problems import gauss_psf
from trips_py_amd import solvers as S

def rate(fn, its, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return its / sorted(ts)[len(ts) // 2]

def leg(name, A, b, L, its):
    for reg in (1e-2, "gcv"):
        out = {}
        for tag, kw in (("one pass", {}), ("two passes", {"fused_orth_iterate": False})):
            r = rate(lambda: S.GKS(A, b, L, 3, its, reg, history=False, **kw), its)
            x, info = S.GKS(A, b, L, 3, its, reg, history=False, **kw)
            out[tag] = (r, x, info)
        d = float(torch.linalg.norm(out["one pass"][1] - out["two passes"][1]) / torch.linalg.norm(out["two passes"][1]))
        print(f"{name} GKS(3, {its}, {reg!r}): one pass {out['one pass'][0]:8.1f} it/s   two passes {out['two passes'][0]:8.1f} it/s   "
              f"|x1 - x2|/|x2| = {d:.2e}   lambda {out['one pass'][2]['regParam']:.6g} / {out['two passes'][2]['regParam']:.6g}", flush=True)

N, nt, na = 256, 32, 15
F = BlockDiagOp([Radon2DParallel(N, np.deg2rad(t + 12.0 * np.arange(na))) for t in range(nt)])
L = SpaceTimeDerivative(N, nt)
x = torch.rand(F.shape[1], device="cuda"); b = F.apply(x)
b = b + 0.01 * torch.randn_like(b) * b.norm() / b.numel() ** 0.5
leg("C5 (32 x 256^2)", F, b, L, 50)
for M in (2048, 4096):
    A = Blur2D(gauss_psf((9, 9), (3, 3))[0], M, M)
    Ld = FirstDerivative2D(M, engine=A.engine)
    xt = torch.rand(M * M, device="cuda"); bb = A.apply(xt)
    bb = bb + 0.01 * torch.randn_like(bb) * bb.norm() / bb.numel() ** 0.5
    leg(f"blur {M}^2", A, bb, Ld, 30)
