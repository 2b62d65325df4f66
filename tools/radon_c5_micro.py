"""C5-shaped projector timing: 32 frames of 256^2, 15 angles per frame, one dynamic handle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trips_py_amd.operators import BlockDiagOp, Radon2DParallel
N, nt, na = 256, int(sys.argv[1]) if len(sys.argv) > 1 else 32, 15
F = BlockDiagOp([Radon2DParallel(N, np.deg2rad(t + 12.0 * np.arange(na))) for t in range(nt)])
x = torch.rand(F.shape[1], device="cuda"); y = torch.empty(F.shape[0], device="cuda"); z = torch.empty(F.shape[1], device="cuda")
for name, fn in (("fwd", lambda: F.apply(x, out=y)), ("adj", lambda: F.apply(y, out=z, transpose=True))):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"c5 {nt} frames {name}: {e0.elapsed_time(e1) / 50 * 1e3:8.1f} us")
