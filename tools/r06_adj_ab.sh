#!/bin/bash
# adjoint tile kernel: 8 against 16 angles per batch at C5's shape and at 1024^2 (round 6)
cd $GRAFT_REPO_ROOT
for env in "" "TRK_RADON_ADJ_AB=16"; do
  echo "== $env"
  env $env python3 tools/radon_c5_micro.py 32 2>&1 | grep "c5"
  env $env python3 tools/radon_c5_micro.py 4 2>&1 | grep "c5"
  env $env python3 tools/radon_micro.py 1024 2>&1 | grep "adj"
done
