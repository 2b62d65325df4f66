#!/bin/bash
# Round 4 GPU visit.  usage: tools/r04_round.sh <out-subdir> [steps...]   steps: test smoke drv prof parity bench
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1; shift
STEPS=${@:-test drv prof parity}
mkdir -p $O
export TMPDIR=/tmp
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
for s in $STEPS; do case $s in
test)
  timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log ;;
smoke)
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log ;;
drv)
  for i in 1 2 3 4 5; do
    timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/drv_$i.json 2> $O/drv_$i.err
    python3 -c "
import json
r = json.load(open('$O/drv_$i.json'))
print('drv $i:', r['value'], {k: r['roofline'][k] for k in ('frac', 'avg_kernel_us', 'median_kernel_us', 'p90_kernel_us', 'min_kernel_us', 'max_kernel_us')})"
  done ;;
bench)
  timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err; echo "bench rc=$?"; cat $O/bench_driver_flags.json ;;
prof)
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_drv -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/prof_drv.json 2> $O/prof_drv.err); echo "prof drv rc=$?"
  f=$(ls -t $O/prof_drv/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/bench_driver_flags_kernel_stats.csv
  cat $O/prof_drv.json | head -c 1200; echo
  head -8 $O/bench_driver_flags_kernel_stats.csv ;;
bench2)
  TRK_DIST_BACKEND=gloo TRK_SINGLE_DEVICE=1 timeout 900 python bench.py --gpus 2 --steps 50 --no-cpu-baseline > $O/bench_2ranks_one_gpu_gloo.json 2> $O/bench2.err; echo "bench2 rc=$?"; cat $O/bench_2ranks_one_gpu_gloo.json; tail -3 $O/bench2.err ;;
parity)
  timeout 900 python3 tools/configs_parity.py > $O/configs_parity.txt 2>&1; echo "parity rc=$?"; cut -c1-400 $O/configs_parity.txt ;;
esac; done
find $O -name "*kernel_trace.csv" -delete 2>/dev/null
find $O -name "*.db" -delete 2>/dev/null
du -sh $O | tail -1
