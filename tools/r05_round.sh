#!/bin/bash
# Round 5 GPU visit.  usage: tools/r05_round.sh <out-subdir> [steps...]
# steps: test (whole -m gpu suite, bars logged) | tnew (this round's new tests) | instr (C3 float64 instrument) | smoke | drv | bench | prof |
#        bench2/bench4/bench8 (ranks on one GPU over gloo) | parity
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1; shift
STEPS=${@:-test instr}
mkdir -p $O
export TMPDIR=/tmp
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
for s in $STEPS; do case $s in
test)
  rm -f $O/bars.txt
  TRK_BARS_LOG=$O/bars.txt timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log ;;
tnew)
  rm -f $O/bars_new.txt
  TRK_BARS_LOG=$O/bars_new.txt timeout 1500 python -m pytest tests/test_gpu_ref64.py tests/test_gpu_solvers.py tests/test_gpu_dist.py tests/test_gpu_kernels.py -m gpu -q > $O/pytest_new.log 2>&1; echo "pytest new rc=$?"; tail -15 $O/pytest_new.log ;;
instr)
  timeout 1200 python3 tools/r05_c3_instrument.py 100 > $O/c3_instrument.txt 2> $O/c3_instrument.err; echo "instr rc=$?"; head -12 $O/c3_instrument.txt; tail -4 $O/c3_instrument.txt; tail -3 $O/c3_instrument.err ;;
smoke)
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log ;;
drv)
  for i in 1 2 3; do
    timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/drv_$i.json 2> $O/drv_$i.err
    python3 -c "
import json
r = json.load(open('$O/drv_$i.json'))
print('drv $i:', r['value'], {k: r['roofline'][k] for k in ('frac', 'avg_kernel_us', 'median_kernel_us', 'min_kernel_us', 'max_kernel_us')})"
  done ;;
bench)
  timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err; echo "bench rc=$?"; cat $O/bench_driver_flags.json ;;
prof)
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_drv -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/prof_drv.json 2> $O/prof_drv.err); echo "prof drv rc=$?"
  f=$(ls -t $O/prof_drv/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/bench_driver_flags_kernel_stats.csv
  head -c 1200 $O/prof_drv.json; echo
  head -8 $O/bench_driver_flags_kernel_stats.csv ;;
bench2|bench4|bench8)
  n=${s#bench}
  TRK_DIST_BACKEND=gloo TRK_SINGLE_DEVICE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus $n --steps 50 --no-cpu-baseline > $O/bench_${n}ranks_one_gpu_gloo.json 2> $O/bench$n.err; echo "bench$n rc=$?"; cat $O/bench_${n}ranks_one_gpu_gloo.json; tail -3 $O/bench$n.err ;;
parity)
  timeout 900 python3 tools/configs_parity.py > $O/configs_parity.txt 2>&1; echo "parity rc=$?"; cut -c1-400 $O/configs_parity.txt ;;
esac; done
find $O -name "*kernel_trace.csv" -delete 2>/dev/null
find $O -name "*.db" -delete 2>/dev/null
du -sh $O | tail -1
