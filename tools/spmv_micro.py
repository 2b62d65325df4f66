"""CSR SpMV (csrc/spmv.hip) alone: the block-diagonal Joseph matrix of bench.py's next_sparse_dynamic leg and the regulariser matrices
(first differences, framelets) — time per apply, GB/s of 8 nnz + 4 (m + n).  TRK_CSR_GROUP=<2..64> forces the lanes per row."""
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
from trips_py_amd.operators import SparseBlockDiag, SparseOp, create_framelet_operator  # noqa: E402
from oracle import cpu_ref as O  # noqa: E402  (matrices only)


def time_op(name, D):
    dev = D.engine.device
    m, n, nnz = D.shape[0], D.shape[1], int(D.matrix.nnz)
    x = torch.rand(n, device=dev)
    y, z = torch.empty(m, device=dev), torch.empty(n, device=dev)
    alg = 8.0 * nnz + 4.0 * (m + n)
    row = [f"{name}: {m} x {n}, nnz {nnz} (mean row {nnz / m:.1f} / column {nnz / n:.1f})"]
    for tag, fn in (("fwd", lambda: D.apply(x, out=y)), ("adj", lambda: D.apply(y, out=z, transpose=True))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        row.append(f"{tag} {us:8.1f} us = {alg / us * 1e-3:7.1f} GB/s ({alg / us * 1e-3 / 8000:.3f} of peak)")
    print("   ".join(row), flush=True)


only = sys.argv[1] if len(sys.argv) > 1 else None       # "joseph": the 17.7 M non-zero matrix alone (for rocprofv3 runs)
T, Nf, na, nd = 16, 256, 10, 256
time_op("joseph 16 x 256^2", SparseBlockDiag(bench.joseph_block_matrix(Nf, [np.deg2rad(t + 18.0 * np.arange(na)) for t in range(T)], nd)))
if only == "joseph":
    sys.exit(0)
time_op("crossphantom-like 16 x 128^2", SparseBlockDiag(bench.joseph_block_matrix(128, [np.deg2rad(t + 36.0 * np.arange(5)) for t in range(16)], 140)))
time_op("first differences 2048^2", SparseOp(O.first_derivative_2d(2048, 2048)))
time_op("space-time differences 16 x 256^2", SparseOp(O.spacetime_derivative(256, 256, 16)))
W = create_framelet_operator(512, 512, 2)
time_op("framelets 512^2 level 2", W)
