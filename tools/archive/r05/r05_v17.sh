#!/bin/bash
# round 5: k_wgram_tv_lds knobs: rows per band, workgroups per CU at one tile of vectors
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/v17; mkdir -p $O; export TMPDIR=/tmp; cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
export TRK_WGRAM_TV_PIECES=2
for band in 32 64 128 256; do
  echo "== LDS_BAND=$band"
  KS=16,24,32 TRK_WGRAM_TV_LDS_BAND=$band timeout 300 python3 tools/wgram_tv_micro.py 2>&1 | grep "^k=" | cut -c1-6,50-60,100-200
done 2>&1 | tee $O/wgram_lds_band.txt
for pc in 1 2; do
  echo "== LDS_PER_CU=$pc (one tile)"
  KS=4,8,12,16 TRK_WGRAM_TV_LDS_PER_CU=$pc timeout 300 python3 tools/wgram_tv_micro.py 2>&1 | grep "^k=" | cut -c1-6,50-60,100-200
done 2>&1 | tee $O/wgram_lds_pc.txt
echo "== register-fed, one tile"; KS=4,8,12,16 TRK_WGRAM_TV_LDS=0 timeout 300 python3 tools/wgram_tv_micro.py 2>&1 | grep "^k=" | cut -c1-6,50-60,100-200
