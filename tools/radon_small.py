"""Projector applies at a small size (default 512^2 x 180): per-kernel breakdown under rocprofv3 (tools/gpu_prof_cmd.sh)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
if os.environ.get("TRK_EXPERIMENT_LIB"):          # a library built with experiment macros (tools/r05_band_exp.sh): timing only
    from trips_py_amd import _lib as _L
    _L.LIB_PATH = os.environ["TRK_EXPERIMENT_LIB"]
    _L._stale = lambda: False
from trips_py_amd.operators import Radon2DParallel
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
R = Radon2DParallel(N, np.linspace(0, np.pi, 180, endpoint=False))
x = torch.rand(N * N, device="cuda"); y = torch.empty(R.shape[0], device="cuda"); z = torch.empty(N * N, device="cuda")
for name, fn in (("fwd", lambda: R.apply(x, out=y)), ("adj", lambda: R.apply(y, out=z, transpose=True))):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"radon {N}^2 x 180 {name}: {e0.elapsed_time(e1) / 50 * 1e3:8.1f} us per apply")
