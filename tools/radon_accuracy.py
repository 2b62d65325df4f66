import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from oracle import cpu_ref as O
from trips_py_amd.operators import Radon2DParallel
for N in (512, 2048, 4096):
    ang = np.array([0.3, 1.1, 2.0])
    R, Ro = Radon2DParallel(N, ang), O.Radon2D(N, ang)
    rng = np.random.default_rng(0)
    ii, jj = np.meshgrid(np.arange(N), np.arange(N), indexing="ij")
    x = (np.exp(-((ii - N / 2.5) ** 2 + (jj - N / 1.7) ** 2) / (0.02 * N * N)) + 0.05 * rng.random((N, N))).reshape(-1)
    xf = x.astype(np.float32).astype(np.float64)
    y, yo = R @ x, Ro @ xf
    print(N, "fwd relerr", np.linalg.norm(y - yo) / np.linalg.norm(yo), "max abs/ max", np.abs(y - yo).max() / np.abs(yo).max())
    s = rng.standard_normal(Ro.shape[0]); sf = s.astype(np.float32).astype(np.float64)
    z, zo = R.T @ s, Ro.T @ sf
    print(N, "adj relerr", np.linalg.norm(z - zo) / np.linalg.norm(zo))
