"""The adjoint at 512^2 x 180 with the parts of a tile as groups of one workgroup (default) or as workgroups (TRK_RADON_ADJ_GROUPS=0):
prints a digest of the output bits (the two must agree: same angle ranges, same order of the partial sums) and the time per apply."""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trips_py_amd.operators import Radon2DParallel
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
na = int(sys.argv[2]) if len(sys.argv) > 2 else 180
R = Radon2DParallel(N, np.linspace(0, np.pi, na, endpoint=False))
g = torch.Generator(device="cuda").manual_seed(5)
y = torch.randn(R.shape[0], device="cuda", generator=g)
z = torch.empty(N * N, device="cuda")
R.apply(y, out=z, transpose=True)
torch.cuda.synchronize()
print("digest", hashlib.sha256(z.cpu().numpy().tobytes()).hexdigest()[:16], "norm", float(z.double().norm()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    R.apply(y, out=z, transpose=True)
e1.record(); torch.cuda.synchronize()
print(f"adjoint {N}^2 x {na}: {e0.elapsed_time(e1) / 50 * 1e3:8.1f} us per apply")
