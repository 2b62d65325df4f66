#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solvers.py tests/test_gpu_core_abi.py tests/test_gpu_blur.py tests/test_gpu_operators.py -m gpu -x -q 2>&1 | tail -4
for i in 1 2 3; do python3 tools/c4_rate.py 2>/dev/null | tail -1; done
