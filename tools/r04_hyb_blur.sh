#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_solvers.py tests/test_gpu_history.py tests/test_gpu_edges.py -m gpu -x -q 2>&1 | tail -3
python3 tools/hybrid_lsqr_blur_rates.py 2>/dev/null
