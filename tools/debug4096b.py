import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trips_py_amd.operators import Blur2D
from trips_py_amd.problems import gauss_psf
dev = torch.device("cuda")
N = 4096
psf, _ = gauss_psf((9, 9), (3, 3))
A = Blur2D(psf, N, N); eng = A.engine
x = torch.randn(N * N, device=dev)
S = eng.scalars(2)
y = A.apply(x)
for trial in range(3):
    y2 = torch.full_like(y, -777.0)
    A.apply(x, out=y2, sumsq=S[0:1])
    torch.cuda.synchronize()
    d = (y2 - y).reshape(N, N)
    bad = (d != 0)
    rows = bad.any(dim=1).nonzero().flatten()
    cols = bad.any(dim=0).nonzero().flatten()
    print("trial", trial, "bad elems", int(bad.sum()), "untouched", int((y2 == -777.0).sum()),
          "rows", rows[:10].tolist(), "...", rows[-5:].tolist(), "n rows", len(rows), "cols", cols[:6].tolist(), cols[-4:].tolist(), len(cols))
    if len(rows):
        r = int(rows[0]); cs = bad[r].nonzero().flatten()
        print("   row", r, "bad cols", cs[:8].tolist(), cs[-4:].tolist(), len(cs), "vals", y2.reshape(N, N)[r, cs[:4]].tolist(), y.reshape(N, N)[r, cs[:4]].tolist())
        # does the bad row equal some other row of the correct output?
        yr = y2.reshape(N, N)[r]
        for dr in range(-40, 41):
            if 0 <= r + dr < N and dr != 0 and torch.equal(yr, y.reshape(N, N)[r + dr]):
                print("   equals correct row", r + dr)
