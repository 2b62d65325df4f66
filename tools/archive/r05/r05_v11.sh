#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/v11; mkdir -p $O; export TMPDIR=/tmp; cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
TRK_BARS_LOG=$O/bars.txt timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_dist.py -m gpu -q -k "wgram or sharded_hip" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
for m in auto 2; do
  echo "== C4 rate, TRK_WGRAM_TV_PIECES=$m"
  if [ $m = auto ]; then timeout 300 python3 tools/c4_rate.py 2>/dev/null | tail -1; else TRK_WGRAM_TV_PIECES=$m timeout 300 python3 tools/c4_rate.py 2>/dev/null | tail -1; fi
done | tee $O/c4_rates.txt
echo "== wgram micro auto"; KS=8,16,24,32 timeout 300 python3 tools/wgram_tv_micro.py 4096 2>/dev/null | cut -c1-200 | tee $O/wgram_micro_auto.txt
bash tools/r05_v10.sh
