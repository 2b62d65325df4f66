import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import cpu_ref as O
from trips_py_amd.operators import Radon2DParallel
N, na, nd = 32, 12, 32
ang = np.linspace(0, np.pi, na, endpoint=False)
R, Ro = Radon2DParallel(N, ang, n_det=nd), O.Radon2D(N, ang, n_det=nd)
rng = np.random.default_rng(0)
x = rng.random(N * N)
x32 = x.astype(np.float32).astype(np.float64)
a = (R @ x).reshape(na, nd); b = (Ro @ x32).reshape(na, nd)
d = np.abs(a - b)
print("max abs diff", d.max(), "rel", np.linalg.norm(a - b) / np.linalg.norm(b))
idx = np.argwhere(d > 1e-6)
print(idx[:20].tolist())
for (ai, di) in idx[:6]:
    print(ai, di, a[ai, di], b[ai, di], "angle deg", np.rad2deg(ang[ai]))
# which image pixel explains the difference? test with unit images at corners
for (i, j) in ((0, 0), (0, N - 1), (N - 1, 0), (N - 1, N - 1), (0, 5), (N - 1, 7)):
    e = np.zeros(N * N); e[i * N + j] = 1.0
    da = np.abs((R @ e) - (Ro @ e)).max()
    print("pixel", i, j, "max diff", da)
