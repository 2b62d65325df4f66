#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/v10; mkdir -p $O; export TMPDIR=/tmp; cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
for ls in -1 1 0; do for pc in 0 2 3; do
  echo "== TRK_WGRAM_TV_LOCKSTEP=$ls TRK_WGRAM_TV_PER_CU=$pc (two pieces)"
  TRK_WGRAM_TV_PIECES=2 TRK_WGRAM_TV_LOCKSTEP=$ls TRK_WGRAM_TV_PER_CU=$pc KS=12,16,17,20,24,25,28 timeout 300 python3 tools/wgram_tv_micro.py 4096 2>/dev/null | cut -c1-200
done; done | tee $O/wgram_lockstep.txt
