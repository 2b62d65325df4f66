#!/bin/bash
# forward quad kernel at 4096^2 x 180 under its knobs (round 6, after the B32 hoist)
cd $GRAFT_REPO_ROOT
for env in "" "TRK_RADON_QBUF=3" "TRK_RADON_BAND=64" "TRK_RADON_BAND=256" "TRK_RADON_QBUF=3 TRK_RADON_BAND=256"; do
  echo "== $env"
  env $env python3 tools/radon_micro.py 4096 2>&1 | grep radon
done
