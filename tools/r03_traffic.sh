#!/bin/bash
# HBM-side bytes per launch (separate FETCH_SIZE / WRITE_SIZE passes) of the headline loop and of C4.  usage: tools/r03_traffic.sh <out-subdir>
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_bench_$ctr -- python3 $R/bench.py --steps 20 --warmup 5 --run-in 50 --no-cpu-baseline --no-extras > /dev/null 2>&1; echo "bench $ctr rc=$?"
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_c4_$ctr -- python3 $R/tools/c4_rate.py > /dev/null 2>&1; echo "c4 $ctr rc=$?"
done
python3 $R/tools/traffic_summary.py $O > $O/traffic.txt 2>&1
cut -c1-175 $O/traffic.txt | grep -v "n=    [0-9] " | head -40
find $O -name "*.csv" -delete; find $O -name "*.db" -delete
