"""CSR SpMV on the 16-frame Joseph block-diagonal matrix (bench.py's next_sparse_dynamic), COLD: three handles holding copies of the matrix
take turns, so every apply streams its 170 MB from HBM.  TRK_CSR_GROUP=<lanes per row> forces the group size (read when the handle is made)."""
import importlib.util
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402
spec = importlib.util.spec_from_file_location("bench", os.path.join(REPO, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
from trips_py_amd.operators import SparseBlockDiag  # noqa: E402

T, Nf, na, nd = 16, 256, 10, 256
blocks = bench.joseph_block_matrix(Nf, [np.deg2rad(t + 18.0 * np.arange(na)) for t in range(T)], nd)
Ds = [SparseBlockDiag(blocks) for _ in range(3)]
dev = Ds[0].engine.device
m, n, nnz = Ds[0].shape[0], Ds[0].shape[1], int(Ds[0].matrix.nnz)
xs = [torch.rand(n, device=dev) for _ in range(3)]
ys = [torch.empty(m, device=dev) for _ in range(3)]
zs = [torch.empty(n, device=dev) for _ in range(3)]
alg = 8.0 * nnz + 4.0 * (m + n)
row = [f"TRK_CSR_GROUP={os.environ.get('TRK_CSR_GROUP', 'rule')}"]
for name in ("fwd", "adj"):
    for mode, nh in (("cold", 3), ("warm", 1)):
        def fn(i):
            h = i % nh
            if name == "fwd":
                Ds[h].apply(xs[h], out=ys[h])
            else:
                Ds[h].apply(ys[h], out=zs[h], transpose=True)
        for i in range(6):
            fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(30):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 30 * 1e3
        row.append(f"{name} {mode} {us:6.1f} us ({alg / us * 1e-3 / 8000:.3f})")
print("   ".join(row), flush=True)
