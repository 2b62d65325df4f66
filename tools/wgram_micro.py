import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from trips_py_amd.engine import default_engine
eng = default_engine()
for m in (16_777_216, 33_546_240):
    kmax = 34
    W = torch.randn((kmax, m), device="cuda"); w = torch.rand(m, device="cuda") + 0.5; b = torch.randn(m, device="cuda")
    G = eng.scalars(kmax * kmax + 2 * kmax)
    for k in (3, 8, 14, 15, 22, 30, 33):
        f = lambda: eng.wgram(W, k, w, b, G.ref(0), G.ref(k * k), G.ref(k * k + k))
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        gb = 4.0 * m * (k + 2) / 1e9
        print(f"m={m} k={k:2d} KA={k+2:2d}: {ms*1e3:8.1f} us  {gb/ms:6.2f} TB/s")
