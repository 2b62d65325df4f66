#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/v12; mkdir -p $O; export TMPDIR=/tmp; cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
TRK_BARS_LOG=$O/bars.txt timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solvers.py -m gpu -q -k "wgram or mmgks or MMGKS" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
grep "auto" $O/bars.txt
for rep in 1 2; do for m in auto 2; do
  echo "== C4 rate, TRK_WGRAM_TV_PIECES=$m"
  if [ $m = auto ]; then timeout 300 python3 tools/c4_rate.py 2>/dev/null | tail -1; else TRK_WGRAM_TV_PIECES=$m timeout 300 python3 tools/c4_rate.py 2>/dev/null | tail -1; fi
done; done | tee $O/c4_rates.txt
echo "== wgram micro auto"; KS=8,16,24,32 timeout 300 python3 tools/wgram_tv_micro.py 4096 2>/dev/null | cut -c1-200 | tee $O/wgram_micro_auto.txt
echo "== wgram micro 2"; TRK_WGRAM_TV_PIECES=2 KS=8,16,24,32 timeout 300 python3 tools/wgram_tv_micro.py 4096 2>/dev/null | cut -c1-200 | tee $O/wgram_micro_p2.txt
