#!/bin/bash
# non-temporal hints on the basis rows of the tall-skinny kernels (TRK_NT bits 6, 7) below the size they start at by rule (11.5 M floats per vector)
R=$GRAFT_REPO_ROOT; cd $R
for nt in -1 192; do
  echo "== TRK_NT=$nt"
  TRK_NT=$nt python3 tools/r06_gcv_gap.py 2>/dev/null | grep "0.01"
  TRK_NT=$nt python3 tools/c3_rates.py 2>/dev/null | tail -3
done
