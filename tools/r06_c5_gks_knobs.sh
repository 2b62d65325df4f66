#!/bin/bash
# C5 GKS (32 frames x 256^2, projection_dim 3, 50 iterations) under the cache-hint mask and the gemv_n launch knobs
R=$GRAFT_REPO_ROOT; cd $R
run() { echo -n "$1: "; env $1 python3 tools/c5_gks_rate.py 32 2>/dev/null | tail -1; }
run "TRK_NT=-1"
run "TRK_NT=192"
run "TRK_NT=195"
run "TRK_GEMVN_GRID=4"
run "TRK_GEMVN_GRID=16"
run "TRK_GEMVN_UNROLL=4"
run "TRK_GEMVN_UNROLL=16"
run "TRK_GEMVT_PER_CU=4"
run "TRK_GEMVT_PER_CU=8"
