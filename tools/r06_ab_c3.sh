#!/bin/bash
# C3's three kernels under the A/B libraries of tools/r06_ab_libs.sh (round 6): rocprofv3 kernel stats of tools/c3_fixed_trace.py
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
for v in product nopipe nof64 flush16 r5like; do
  if [ $v = product ]; then unset TRK_EXPERIMENT_LIB; else export TRK_EXPERIMENT_LIB=$R/tools/experiments/lib/libtrk_$v.so; fi
  rm -rf /tmp/ab_$v
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$v -- python3 $R/tools/c3_fixed_trace.py > /tmp/ab_$v.log 2>&1
  f=$(ls -t /tmp/ab_$v/*/*kernel_stats.csv | head -1)
  echo "== $v"; python3 $R/tools/stats_top.py $f | head -3 | cut -c1-110
done
unset TRK_EXPERIMENT_LIB
cd $R
for v in product nopipe nof64 flush16 r5like; do
  if [ $v = product ]; then unset TRK_EXPERIMENT_LIB; else export TRK_EXPERIMENT_LIB=$R/tools/experiments/lib/libtrk_$v.so; fi
  echo "== $v rates"; python3 tools/c3_rates.py 2>/dev/null | tail -1
done
