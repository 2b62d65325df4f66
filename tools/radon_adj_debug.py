"""Debug aid: the two adjoint kernels against the float64 oracle and each other (GPU box; the oracle is the checker)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import cpu_ref as O
from trips_py_amd.operators import Radon2DParallel
for N, ang in ((64, [0.3]), (64, [1.1]), (64, [0.3, 1.1, 2.0]), (96, list(np.linspace(0, np.pi, 9, endpoint=False))), (512, [0.3, 1.1, 2.0])):
    ang = np.asarray(ang)
    R, Ro = Radon2DParallel(N, ang), O.Radon2D(N, ang)
    rng = np.random.default_rng(0)
    y = rng.standard_normal(Ro.shape[0]).astype(np.float32).astype(np.float64)
    ref = Ro.T @ y
    yd = torch.from_numpy(y.astype(np.float32)).cuda()
    t = R.apply(yd, transpose=True).cpu().numpy().astype(np.float64)
    os.environ["TRK_RADON_ADJ_SIMPLE"] = "1"
    s = R.apply(yd, transpose=True).cpu().numpy().astype(np.float64)
    del os.environ["TRK_RADON_ADJ_SIMPLE"]
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
    print(f"N={N} na={len(ang)}: simple vs oracle {rel(s, ref):.2e}  tile vs oracle {rel(t, ref):.2e}  tile vs simple {rel(t, s):.2e}")
    if rel(t, s) > 1e-6:
        d = np.abs(t - s).reshape(N, N)
        bad = np.argwhere(d > 1e-6 * np.abs(s).max())
        print("   differing pixels:", len(bad), "first", bad[:6].tolist(), "rows", np.unique(bad[:, 0])[:10], "cols", np.unique(bad[:, 1])[:10])
