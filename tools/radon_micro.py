import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trips_py_amd.operators import Radon2DParallel
sizes = [int(a) for a in sys.argv[1:]] or [256, 512, 1024, 2048, 4096]
for N in sizes:
    na = 180
    R = Radon2DParallel(N, np.linspace(0, np.pi, na, endpoint=False))
    x = torch.rand(N * N, device="cuda"); y = torch.empty(R.shape[0], device="cuda"); z = torch.empty(N * N, device="cuda")
    for name, fn in (("fwd", lambda: R.apply(x, out=y)), ("adj", lambda: R.apply(y, out=z, transpose=True))):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20 if N <= 2048 else 5
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        taps = 2.0 * N * N * na
        print(f"radon {N}^2 x {na} {name}: {ms:9.3f} ms  {taps/ms/1e9:8.2f} Ttaps/s  alg {4*(N*N+na*N)/ms/1e6:8.1f} GB/s")
