"""The reference's DEFAULT regparam is 'gcv' in GKS / MMGKS / the hybrid solvers: rates with 'gcv' against a number, on C5's shape (GKS) and
at 2048^2 / 4096^2 (MMGKS with TV), same calls as bench.py's."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trips_py_amd.operators import Radon2DParallel, BlockDiagOp, SpaceTimeDerivative, Blur2D, FirstDerivative2D
from trips_py_amd.problems import gauss_psf
from trips_py_amd import solvers as S

def rate(fn, its, reps=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return its / sorted(ts)[len(ts) // 2]

N, nt, na = 256, 32, 15
F = BlockDiagOp([Radon2DParallel(N, np.deg2rad(t * 3.0 + 12.0 * np.arange(na))) for t in range(nt)])
L = SpaceTimeDerivative(N, nt)
x = torch.rand(F.shape[1], device="cuda"); b = F.apply(x)
for reg in (1e-2, "gcv"):
    print(f"C5 GKS(projection_dim=3, n_iter=50, regparam={reg!r}): {rate(lambda: S.GKS(F, b, L, 3, 50, reg, history=False), 50):8.0f} it/s")
for M in (2048, 4096):
    A = Blur2D(gauss_psf((9, 9), (3, 3))[0], M, M)
    Ld = FirstDerivative2D(M, engine=A.engine)
    xt = torch.rand(M * M, device="cuda"); bb = A.apply(xt)
    bb = bb + 0.01 * torch.randn_like(bb) * bb.norm() / bb.numel() ** 0.5
    for reg in (1e-2, "gcv"):
        r = rate(lambda: S.MMGKS(A, bb, Ld, pnorm=2, qnorm=1, projection_dim=3, n_iter=30, regparam=reg, epsilon=0.1, history=False), 30)
        print(f"MMGKS TV {M}^2 (n_iter=30, regparam={reg!r}): {r:8.1f} it/s")
