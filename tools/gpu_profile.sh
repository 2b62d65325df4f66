#!/bin/bash
# bench + kernel-trace stats + PMC traffic for the dominant kernel -> gpurun_out/profile_r01/
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/profile_r01
mkdir -p $O
export TMPDIR=/tmp
cd $R
python bench.py --steps 100 --warmup 10 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cat $O/bench.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras > $O/trace_bench.json 2> $O/trace.err; echo "trace rc=$?"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2>&1; echo "pmc1 rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2>&1; echo "pmc2 rc=$?"
cd $R
python3 tools/profile_summary.py $O > $O/summary.txt; cat $O/summary.txt
