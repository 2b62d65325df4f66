#!/bin/bash
# round 5: the LDS-staged TV Gram (k_wgram_tv_lds): parity, then time against the register-fed kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/v16; mkdir -p $O; export TMPDIR=/tmp; cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "wgram_tv" 2>&1 | tail -15
export KS=${KS:-8,16,20,24,32}
for pcs in 2 1; do
for lds in 1 0; do
  echo "== TRK_WGRAM_TV_LDS=$lds  pieces/auto=$pcs"
  if [ $pcs = 2 ]; then export TRK_WGRAM_TV_PIECES=2; else unset TRK_WGRAM_TV_PIECES; fi
  TRK_WGRAM_TV_LDS=$lds timeout 300 python3 tools/wgram_tv_micro.py 2>&1 | grep "^k=" | cut -c1-60,100-200
done; done 2>&1 | tee $O/wgram_lds.txt
