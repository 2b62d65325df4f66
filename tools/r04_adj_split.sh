#!/bin/bash
# Adjoint with a tile's angles split over 1 / 2 / 4 / 8 workgroups (TRK_RADON_ADJ_SPLIT; default: 8 up to 128 tiles, else 4): time per
# apply at small sizes, then the tests.
R=$GRAFT_REPO_ROOT; cd $R
for N in 256 512 768 1024; do
  for sp in 1 2 4 8; do
    echo "N=$N split=$sp: $(TRK_RADON_ADJ_SPLIT=$sp python3 tools/radon_small.py $N 2>/dev/null | grep adj | tr '\n' ' ')"
  done
done
timeout 900 python -m pytest tests/test_gpu_radon_accuracy.py tests/test_gpu_operators.py tests/test_gpu_solvers.py -m gpu -x -q 2>&1 | tail -8
python3 tools/c3_rates.py 2>&1 | tail -3
