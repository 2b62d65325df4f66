#!/bin/bash
# GPU visit 2 of round 5: the instrument with emulated partial sums, the adversarial Gram test with two / three pieces, Gram timings
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/v2; mkdir -p $O; export TMPDIR=/tmp; cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
TRK_BARS_LOG=$O/bars_a.txt timeout 900 python -m pytest tests/test_gpu_ref64.py tests/test_gpu_kernels.py tests/test_gpu_operators.py -m gpu -q -k "ref64 or float64 or wgram or fan or table or arithmetic or half_step" > $O/pytest_a.log 2>&1; echo "pytest a rc=$?"; tail -8 $O/pytest_a.log
TRK_WGRAM_TV_PIECES=2 TRK_BARS_LOG=$O/bars_2piece.txt timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "adversarial" > $O/pytest_2piece.log 2>&1; echo "pytest 2piece rc=$?"; tail -3 $O/pytest_2piece.log
cat $O/bars_2piece.txt; grep wgram $O/bars_a.txt
for v in "3 0" "2 0" "3 1"; do set -- $v
  echo "== pieces $1 occ2 $2"; TRK_WGRAM_TV_PIECES=$1 TRK_WGRAM_TV_OCC2=$2 KS=8,16,17,20,24,25,32,33 timeout 600 python3 tools/wgram_tv_micro.py 4096 2>/dev/null | tee $O/wgram_micro_p$1_o$2.txt
done
timeout 1500 python3 tools/r05_c3_instrument.py 60 > $O/c3_instrument.txt 2> $O/c3_instrument.err; echo "instr rc=$?"; grep "^#" $O/c3_instrument.txt; tail -3 $O/c3_instrument.err
