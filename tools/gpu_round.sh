#!/bin/bash
# One GPU-box visit: tests, default bench, 2-rank bench on one GPU (gloo), kernel-trace stats of C3 / C4 / C5 and the
# FETCH/WRITE passes of C4.  usage: tools/gpu_round.sh <out-subdir> [steps...]   steps: test bench bench2 prof pmc
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1; shift
STEPS=${@:-test bench bench2 prof pmc}
mkdir -p $O
export TMPDIR=/tmp
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
for s in $STEPS; do case $s in
test)
  timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log ;;
smoke)
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log ;;
bench)
  timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cat $O/bench.json ;;
bench2)
  TRK_DIST_BACKEND=gloo TRK_SINGLE_DEVICE=1 timeout 900 python bench.py --gpus 2 --steps 50 --no-cpu-baseline > $O/bench_2ranks_gloo.json 2> $O/bench2.err; echo "bench2 rc=$?"; cat $O/bench_2ranks_gloo.json; tail -3 $O/bench2.err ;;
prof)
  for c in c3 c4 c5; do
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$c -- python3 $R/tools/configs_micro.py $c > $O/prof_$c.log 2>&1); echo "prof $c rc=$?"
    f=$(ls -t $O/prof_$c/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/${c}_kernel_stats.csv
    grep -- "->\|ms$" $O/prof_$c.log | head -12
  done
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras > $O/prof_bench.json 2> $O/prof_bench.err); echo "prof bench rc=$?"
  f=$(ls -t $O/prof_bench/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/bench_cgls4096_kernel_stats.csv ;;
pmc)
  for ctr in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_c4_$ctr -- python3 $R/tools/configs_micro.py c4 > $O/pmc_c4_$ctr.log 2>&1); echo "pmc c4 $ctr rc=$?"
    (cd /tmp && timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_bench_$ctr -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2>&1); echo "pmc bench $ctr rc=$?"
  done
  python3 tools/traffic_summary.py $O > $O/traffic_summary.txt 2>&1; head -40 $O/traffic_summary.txt ;;
esac; done
# keep the merge small: raw traces stay on the box
find $O -name "*kernel_trace.csv" -delete 2>/dev/null
find $O -name "*.db" -delete 2>/dev/null
du -sh $O | tail -1
