#!/bin/bash
# TIMING EXPERIMENT (round 6): k_radon_fwd_band with every task carrying a second, mirrored ray on the same weights (a pair of symmetric
# angles sharing one march) — wrong results, the time of the real thing.  Library: tools/r06_ab_libs.sh bandpair=-DTRK_BAND_EXPERIMENT_PAIR
R=$GRAFT_REPO_ROOT; cd $R
echo "== product"; python3 tools/radon_micro.py 256 512 1024 2>/dev/null | grep fwd
echo "== paired tasks"; TRK_EXPERIMENT_LIB=$R/tools/experiments/lib/libtrk_bandpair.so python3 tools/radon_micro.py 256 512 1024 2>/dev/null | grep fwd
