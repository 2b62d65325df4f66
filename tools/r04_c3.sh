#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_operators.py tests/test_gpu_solvers.py tests/test_gpu_core_abi.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2 3; do python3 tools/c3_rates.py 2>/dev/null | tail -3 | tr '\n' ' '; echo; done
python3 tools/c5_cgls_rate.py 2>/dev/null | tail -2
python3 tools/c5_gks_rate.py 2>/dev/null | tail -1
