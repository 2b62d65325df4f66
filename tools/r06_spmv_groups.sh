#!/bin/bash
cd $GRAFT_REPO_ROOT
for g in "" 8 16 32 64; do
  if [ -z "$g" ]; then python3 tools/r06_spmv_cold.py 2>&1 | grep TRK_CSR; else TRK_CSR_GROUP=$g python3 tools/r06_spmv_cold.py 2>&1 | grep TRK_CSR; fi
done
