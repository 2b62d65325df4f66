"""Twenty fan-beam adjoint applies at 512^2 x 180 x 724 (argv[1] = march2: rounds 3-5's kernel) — the program behind the counter passes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "march2":
    os.environ["TRK_FAN_ADJ_MARCH2"] = "1"
import torch
from trips_py_amd.operators import FanBeam2D
N = 512
R = FanBeam2D(N, views=180)
y = torch.randn(R.shape[0], device=R.engine.device)
z = torch.empty(N * N, device=R.engine.device)
for _ in range(20):
    R.apply(y, out=z, transpose=True)
torch.cuda.synchronize()
