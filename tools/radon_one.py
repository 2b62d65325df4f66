import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trips_py_amd.operators import Radon2DParallel
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
R = Radon2DParallel(N, np.linspace(0, np.pi, 180, endpoint=False))
x = torch.rand(N * N, device="cuda"); y = torch.empty(R.shape[0], device="cuda"); z = torch.empty(N * N, device="cuda")
for _ in range(3):
    R.apply(x, out=y); R.apply(y, out=z, transpose=True)
torch.cuda.synchronize()
