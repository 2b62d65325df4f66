#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
tools/gpu_pmc_cmd.sh k_radon_adj_tile tools/radon_small.py 512
cd $R; tools/r04_c3trace.sh 2>&1 | grep -v "^E2026\|^W2026" | head -40
