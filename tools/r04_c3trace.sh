#!/bin/bash
# C3 (fixed-lambda Hybrid-LSQR, 512^2 x 180) behind rocprofv3 --kernel-trace: per-kernel stats and the timeline of the last launches.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/c3trace
export TMPDIR=/tmp
mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python3 $R/tools/c3_fixed_trace.py > $O/run.log 2>&1
tail -3 $O/run.log
f=$(ls -t $O/p/*/*kernel_stats.csv | head -1); python3 $R/tools/stats_top.py $f | head -14
t=$(ls -t $O/p/*/*kernel_trace.csv | head -1); python3 $R/tools/trace_gaps.py $t 40 | tail -60
cd $R; python3 tools/c3_rates.py 2>&1 | tail -8
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
