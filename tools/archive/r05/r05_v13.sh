#!/bin/bash
# round 5: the counters and per-kernel tables behind profiles/r05 — HBM-side traffic of the headline loop and of C4 (separate FETCH_SIZE /
# WRITE_SIZE passes), per-kernel durations of C3 / C4 / C5 (rocprofv3 --kernel-trace --stats over tools/configs_micro.py)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/v13; mkdir -p $O; export TMPDIR=/tmp; cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
cd /tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_bench_$ctr -- python3 $R/bench.py --steps 20 --warmup 5 --run-in 50 --no-cpu-baseline --no-extras > /dev/null 2>&1; echo "bench $ctr rc=$?"
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_c4_$ctr -- python3 $R/tools/c4_rate.py > /dev/null 2>&1; echo "c4 $ctr rc=$?"
done
python3 $R/tools/traffic_summary.py $O > $O/traffic.txt 2>&1
cut -c1-175 $O/traffic.txt | grep -v "n=    [0-9] " | head -30
for c in c3 c4 c5; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$c -- python3 $R/tools/configs_micro.py $c > $O/prof_$c.log 2>&1; echo "prof $c rc=$?"
  f=$(ls -t $O/prof_$c/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/${c}_kernel_stats.csv
  python3 $R/tools/stats_top.py $O/${c}_kernel_stats.csv 2>/dev/null | head -14
done
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; rm -rf $O/prof_c3 $O/prof_c4 $O/prof_c5 $O/pmc_*
du -sh $O
