"""Time the BASELINE configs C2-C5 (single GPU) through the product solvers; used with and without rocprofv3."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trips_py_amd.operators import Blur2D, Radon2DParallel, BlockDiagOp, FirstDerivative2D, SpaceTimeDerivative
from trips_py_amd.problems import gauss_psf
from trips_py_amd import solvers as S

which = sys.argv[1] if len(sys.argv) > 1 else "all"
dev = torch.device("cuda")
psf, _ = gauss_psf((9, 9), (3, 3))


def sync():
    torch.cuda.synchronize()


def image(N, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    img = torch.zeros((N, N))
    rr = torch.randint(0, N - N // 8, (8, 2), generator=g); hw = torch.randint(N // 16, N // 3, (8, 2), generator=g)
    for q in range(8):
        img[rr[q, 0]:rr[q, 0] + hw[q, 0], rr[q, 1]:rr[q, 1] + hw[q, 1]] += 0.2 + 0.1 * q
    return (img + 0.1 * torch.rand((N, N), generator=g)).reshape(-1).to(dev)


def noisy(b, seed=1):
    e = torch.randn(b.numel(), device=dev, generator=torch.Generator(device=dev).manual_seed(seed))
    return b + e * (0.01 * torch.linalg.norm(b) / torch.linalg.norm(e))


def timed(name, fn, reps=2):
    fn(); sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    sync()
    dt = (time.perf_counter() - t0) / reps
    print(f"{name:58s} {dt*1e3:10.2f} ms")
    return out, dt


if which in ("all", "c2"):
    N = 512; A = Blur2D(psf, N, N); xt = image(N); b = noisy(A.apply(xt))
    x0 = torch.zeros(N * N, device=dev)
    (_, info), dt = timed("C2 blur512 CGLS 100 its (history on)", lambda: S.CGLS(A, b, x0, 100, 0))
    print(f"   -> {100/dt:.0f} it/s")
    (_, info), dt = timed("C2 blur512 CGLS 100 its (history off)", lambda: S.CGLS(A, b, x0, 100, 0, history=False))
    print(f"   -> {100/dt:.0f} it/s")
if which in ("all", "c3"):
    N = 512; ang = np.linspace(0, np.pi, 180, endpoint=False)
    R = Radon2DParallel(N, ang); xt = image(N); b = noisy(R.apply(xt))
    y = torch.empty(R.shape[0], device=dev); z = torch.empty(N * N, device=dev)
    _, dt = timed("C3 radon512x180 forward apply", lambda: R.apply(xt, out=y), reps=20)
    print(f"   -> {2*N*N*180/dt/1e9:.1f} Gtaps/s")
    _, dt = timed("C3 radon512x180 adjoint apply", lambda: R.apply(y, out=z, transpose=True), reps=20)
    (_, info), dt = timed("C3 Hybrid_LSQR 100 its lam=1e-2", lambda: S.Hybrid_LSQR(R, b, 100, 1e-2))
    print(f"   -> {100/dt:.1f} it/s")
    (_, info), dt = timed("C3 Hybrid_LSQR 100 its gcv", lambda: S.Hybrid_LSQR(R, b, 100, "gcv"))
    print(f"   -> {100/dt:.1f} it/s")
if which in ("all", "c4"):
    N = 4096; A = Blur2D(psf, N, N); L = FirstDerivative2D(N); xt = image(N); b = noisy(A.apply(xt))
    (_, info), dt = timed("C4 blur4096 MMGKS p=2 q=1 d=3 30 its lam=1e-2 (history off)",
                          lambda: S.MMGKS(A, b, L, 2, 1, 3, 30, 1e-2, history=False), reps=1)
    print(f"   -> {30/dt:.1f} it/s")
    (_, info), dt = timed("C4 blur4096 GKS d=3 30 its lam=1e-2 (history off)",
                          lambda: S.GKS(A, b, L, 3, 30, 1e-2, history=False), reps=1)
    print(f"   -> {30/dt:.1f} it/s")
if which in ("all", "c5"):
    N, nt, na = 256, 32, 15
    ops = [Radon2DParallel(N, np.deg2rad(t + 12.0 * np.arange(na))) for t in range(nt)]
    F = BlockDiagOp(ops); L = SpaceTimeDerivative(N, nt)
    xt = torch.cat([image(N, t) for t in range(nt)]); b = noisy(F.apply(xt))
    (_, info), dt = timed("C5 dyn 32x256^2 (15 ang/frame) GKS d=3 50 its lam=1e-2", lambda: S.GKS(F, b, L, 3, 50, 1e-2, history=False))
    print(f"   -> {50/dt:.1f} it/s")
    x0 = torch.zeros(F.shape[1], device=dev)
    (_, info), dt = timed("C5 dyn 32x256^2 CGLS 100 its", lambda: S.CGLS(F, b, x0, 100, 0, history=False))
    print(f"   -> {100/dt:.0f} it/s")
