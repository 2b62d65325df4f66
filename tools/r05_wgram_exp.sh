#!/bin/bash
# round 5: where the TV Gram pass spends its time — the same kernel (a) without the fetch of the column right of a workgroup's strips,
# (b) with its loads alone.  Builds a SEPARATE library with -DTRK_WGRAM_TV_EXPERIMENT (results are wrong by construction: timing only).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/wexp; mkdir -p $O; export TMPDIR=/tmp; cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
C=$R/trips_py_amd/csrc
hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -DTRK_WGRAM_TV_EXPERIMENT -I$R/include -I$C -c $C/vecops.hip -o /tmp/vecops_exp.o || exit 1
objs=$(ls $C/*.o | grep -v vecops.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libtrk_exp.so /tmp/vecops_exp.o $objs -ldl || exit 1
export TRK_EXPERIMENT_LIB=/tmp/libtrk_exp.so TRK_WGRAM_TV_PIECES=2 KS=${KS:-16,24,32}
for x in 0 4 8 12; do
  echo "== TRK_WGRAM_TV_X=$x  (4: no neighbour-column fetch, 8: loads only)"
  TRK_WGRAM_TV_X=$x python3 tools/wgram_tv_micro.py | cut -c1-60,100-200
done 2>&1 | tee $O/wgram_exp.txt
