#!/bin/bash
R=$GRAFT_REPO_ROOT
for d in 6 9; do for rpb in 0 64; do
  echo "== D=$d RPB=$rpb"; TRK_BLUR_D=$d TRK_BLUR_RPB=$rpb python3 $R/tools/blur_micro.py 4096 30 2>&1 | grep -E "^blur"
done; done
python3 $R/tools/blur_micro.py 512 50 | grep blur
python3 $R/tools/blur_micro.py 2048 50 | grep blur
timeout 600 python -m pytest tests/test_gpu_blur.py tests/test_gpu_fullsize.py tests/test_gpu_cgls.py -m gpu -q 2>&1 | tail -5
