#!/bin/bash
# Hybrid-LSQR with automatic lambda: C3 (512^2 x 180) and the 512^2 / 2048^2 blur, then the interpreter's profile of a dp solve
R=$GRAFT_REPO_ROOT; cd $R
python3 tools/c3_rates.py profile 2>/dev/null | head -45
python3 tools/hybrid_lsqr_blur_rates.py 2>/dev/null
