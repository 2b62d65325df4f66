"""The weighted Gram of L V at 4096^2: trk_wgram over the stored images (2n floats per vector) vs trk_wgram_tv from V (n floats)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if os.environ.get("TRK_EXPERIMENT_LIB"):          # a library built with experiment macros (tools/r05_wgram_exp.sh): timing only
    from trips_py_amd import _lib as _L
    _L.LIB_PATH = os.environ["TRK_EXPERIMENT_LIB"]
    _L._stale = lambda: False
from trips_py_amd.engine import default_engine
from trips_py_amd.operators import FirstDerivative2D
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
eng = default_engine()
L = FirstDerivative2D(N, engine=eng)
n, p = N * N, 2 * N * (N - 1)
kmax = 33
V = torch.randn(kmax, n, device=eng.device)
LV = torch.empty(kmax, p, device=eng.device)
for j in range(kmax):
    L.apply(V[j], out=LV[j])
w = torch.rand(p, device=eng.device) + 0.25
G = eng.scalars(2 * kmax * kmax + kmax)
for k in ([int(v) for v in os.environ["KS"].split(",")] if "KS" in os.environ else (3, 8, 16, 17, 24, 32, 33)):
    out = []
    zz = V[kmax - 1]
    for f in (lambda: eng.wgram(LV, k, w, None, G[0:k * k]), lambda: eng.wgram_tv(V, k, N, w, G[k * k:2 * k * k]),
              lambda: eng.wgram_tv(V, k, N, w, G[k * k:2 * k * k], z=zz, h=G[2 * kmax * kmax - kmax:2 * kmax * kmax - kmax + k])):
        f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            f()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / 10 * 1e6)
    g = eng.to_host(G)
    a, b = g[:k * k], g[k * k:2 * k * k]
    print(f"k={k:2d}: stored images {out[0]:8.1f} us ({(k * p + p) * 4 / out[0] / 1e6:5.2f} TB/s)   from V {out[1]:8.1f} us ({(k * n * 4 + p * 4) / out[1] / 1e6:5.2f} TB/s of its own bytes)   with V^T z {out[2]:8.1f} us   max rel diff {abs(a - b).max() / abs(a).max():.1e}")
