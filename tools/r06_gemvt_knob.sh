#!/bin/bash
# Hybrid-GMRES on the 512^2 blur under TRK_GEMVT_F4 = 1 / 2 / 4 (float4s of a basis row per thread of k_gemv_t / k_gemv_t2 at least)
R=$GRAFT_REPO_ROOT; cd $R
for f in 1 2 4; do echo "TRK_GEMVT_F4=$f"; TRK_GEMVT_F4=$f python3 tools/gmres_rates.py 2>/dev/null | head -2; done
