#!/bin/bash
# CSR SpMV cold / warm on the 16-frame Joseph matrix by workgroups per CU (TRK_CSR_GRID_PER_CU: 16 = the product's cap; the matrix wants 10 per CU
# = 1.25 resident rounds of 8)
R=$GRAFT_REPO_ROOT; cd $R
for g in 16 10 8 5 4; do echo -n "TRK_CSR_GRID_PER_CU=$g   "; TRK_CSR_GRID_PER_CU=$g python3 tools/r06_spmv_cold.py 2>/dev/null | tail -1; done
