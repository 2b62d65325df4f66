#!/bin/bash
# round 5: adjoint knobs at small sizes (angles per batch, split) + forward band kernel slices
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/v26; mkdir -p $O; export TMPDIR=/tmp; cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
for N in 256 512; do
for cfg in "" "TRK_RADON_ADJ_AB=16" "TRK_RADON_ADJ_SPLIT=2" "TRK_RADON_ADJ_SPLIT=8" "TRK_RADON_ADJ_AB=16 TRK_RADON_ADJ_SPLIT=8" "TRK_RADON_ADJ_AB=16 TRK_RADON_ADJ_SPLIT=2"; do
  echo "== N=$N $cfg"; env $cfg python3 tools/radon_small.py $N 2>&1 | grep "radon"
done; done 2>&1 | tee $O/adj_knobs.txt
