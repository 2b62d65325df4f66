import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trips_py_amd.operators import FanBeam2D
for N, views in ((256, 90), (512, 180), (1024, 180)):
    R = FanBeam2D(N, views=views)
    x = torch.rand(N * N, device="cuda"); y = torch.empty(R.shape[0], device="cuda"); z = torch.empty(N * N, device="cuda")
    for name, fn in (("fwd", lambda: R.apply(x, out=y)), ("adj", lambda: R.apply(y, out=z, transpose=True))):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"fanbeam {N}^2 x {views} views x {R.n_det} det {name}: {e0.elapsed_time(e1)/20:9.3f} ms")
