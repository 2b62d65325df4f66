"""Does the 4096-float row pitch (16 KiB, power of two) cost bandwidth?  Blur + copy GB/s over nearby sizes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from trips_py_amd.operators import Blur2D
from trips_py_amd.problems import gauss_psf
K = int(os.environ.get("PSF", "9"))               # PSF side (3, 5, 7, 9: sliding kernel; 11..15: strip kernel)
psf = gauss_psf((K, K), (K / 3.0, K / 3.0))[0]
SIZES = [int(a) for a in sys.argv[1:]] or (1024, 1536, 2048, 2560, 3072, 3584, 3840, 4000, 4096, 4160, 4224, 4352, 5120, 8192)
for N in SIZES:
    A = Blur2D(psf, N, N)
    x = torch.randn(N * N, device="cuda"); y = torch.empty_like(x)
    def t(fn, R=40):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(R): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / R * 1e-3
    tb, tc = t(lambda: A.apply(x, out=y)), t(lambda: y.copy_(x))
    print(f"N={N}: blur {tb*1e6:7.2f} us {8*N*N/tb/1e12:5.2f} TB/s   copy {tc*1e6:7.2f} us {8*N*N/tc/1e12:5.2f} TB/s")
