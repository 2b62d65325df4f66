#!/bin/bash
R=$GRAFT_REPO_ROOT
for d in 3 6 9 18; do for rpb in 46 64 100; do
  echo "== D=$d RPB=$rpb"; TRK_BLUR_D=$d TRK_BLUR_RPB=$rpb python3 $R/tools/blur_micro.py 4096 30 2>&1 | grep -E "^blur"
done; done
