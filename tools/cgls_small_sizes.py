"""CGLS(100 iterations, tol = 0) on small blur problems: the tiled forms (1: four blurs per iteration, 2: two, w by recurrence) vs the
streaming form — iterations/s with history off, and how far each form's iterates are from the streaming form's (x_true given, 1 %
noise: the shape of the C2 configuration)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from trips_py_amd.operators import Blur2D
from trips_py_amd.problems import gauss_psf
from trips_py_amd.solvers import CGLS
psf = gauss_psf((9, 9), (3, 3))[0]
for N in [int(a) for a in sys.argv[1:]] or [128, 256, 512, 768, 1024]:
    A = Blur2D(psf, N, N)
    dev = A.engine.device
    g = torch.Generator(device=dev).manual_seed(3)
    xt = torch.rand(N * N, device=dev, generator=g)
    b = A.apply(xt)
    b = b + 0.01 * torch.randn(N * N, device=dev, generator=g) * b.norm() / N
    x0 = torch.zeros(N * N, device=dev)
    out = []
    for kw in ({"tiled": 1}, {"tiled": 2}, {"tiled": False}):
        CGLS(A, b, x0, 100, 0, history=False, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            CGLS(A, b, x0, 100, 0, history=False, **kw)
        torch.cuda.synchronize()
        out.append(500 / (time.perf_counter() - t0))
    xs, info_s = CGLS(A, b, x0, 100, 0, xt, tiled=False, fused=False)
    dev_of = []
    for tl in (1, 2):
        x, info = CGLS(A, b, x0, 100, 0, xt, tiled=tl)
        d = [float(torch.linalg.norm(h.reshape(-1) - hs.reshape(-1)) / torch.linalg.norm(hs)) for h, hs in zip(info["xHistory"], info_s["xHistory"])]
        dev_of.append((max(d[:25]), max(d)))
    print(f"N={N}: tiled(1) {out[0]:9.0f} it/s   tiled(2) {out[1]:9.0f} it/s   streaming {out[2]:9.0f} it/s | max rel. distance from the "
          f"streaming iterates over 25 / 100 iterations: tiled(1) {dev_of[0][0]:.1e} / {dev_of[0][1]:.1e}, tiled(2) {dev_of[1][0]:.1e} / {dev_of[1][1]:.1e}")
