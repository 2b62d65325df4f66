#!/bin/bash
# the mirrored-pair adjoint under its knob (quads per batch), and C3's rates (round 6)
cd $GRAFT_REPO_ROOT
for env in "" "TRK_RADON_ADJQ_QB=2"; do
  echo "== $env"
  env $env python3 tools/radon_micro.py 4096 2048 2>&1 | grep radon
done
python3 tools/c3_rates.py 2>&1 | tail -8
