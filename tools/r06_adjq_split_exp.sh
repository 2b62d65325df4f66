#!/bin/bash
# TIMING EXPERIMENT (round 6): k_radon_adj_quad at 512^2 / 1024^2 with the quads of an orbit split over blockIdx.z (results wrong — the
# splits overwrite each other — the time is what the real thing would cost before its combining step).  Library: tools/r06_ab_libs.sh qsplit=-DTRK_ADJQ_EXPERIMENT_SPLIT
R=$GRAFT_REPO_ROOT; cd $R
echo "== product (k_radon_adj_tile at 512 / 1024)"; python3 tools/radon_micro.py 512 768 1024 2>/dev/null | grep adj
for sp in 1 2 4 8; do
  echo "== quad kernel, split $sp (TRK_RADON_ADJQ_MIN=512)"
  TRK_EXPERIMENT_LIB=$R/tools/experiments/lib/libtrk_qsplit.so TRK_RADON_ADJQ_MIN=512 TRK_ADJQ_SPLIT=$sp python3 tools/radon_micro.py 512 768 1024 2>/dev/null | grep adj
done
