import sys, json
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for i in range(3):
    r = bench.extra_sparse_dynamic(1)
    print({k: r[k] for k in ('iters_per_sec_all_ranks', 'median_solve_iters_per_sec', 'ms_per_solve')}, flush=True)
