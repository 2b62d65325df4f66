#!/bin/bash
# Forward projector at small sizes against the band height (TRK_RADON_BAND): the per-wave-window kernel holds 4 workgroups per CU
# (120 registers), so a grid of more than 1024 workgroups runs in more than one round.
R=$GRAFT_REPO_ROOT; cd $R
for N in 256 512 768; do
  for b in 64 96 128 176 192 256 384 512; do
    echo "N=$N band=$b: $(TRK_RADON_BAND=$b python3 tools/radon_small.py $N 2>/dev/null | grep fwd | tr '\n' ' ')"
  done
done
for b in 128 176 256; do echo "band=$b: $(TRK_RADON_BAND=$b python3 tools/c3_rates.py 2>/dev/null | tail -1)"; done
