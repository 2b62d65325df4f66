#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/v8; mkdir -p $O; export TMPDIR=/tmp; cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
timeout 300 python3 tools/gmres_rates.py 2>/dev/null | tee $O/gmres_rates.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_spmv -- python3 $R/tools/spmv_micro.py joseph > $O/prof_spmv.log 2>&1; echo "prof rc=$?"
f=$(ls -t $O/prof_spmv/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/spmv_kernel_stats.csv; head -6 $O/spmv_kernel_stats.csv | cut -c1-200
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_spmv_$ctr -- python3 $R/tools/spmv_micro.py joseph > /dev/null 2>&1; echo "pmc $ctr rc=$?"
done
python3 $R/tools/traffic_summary.py $O > $O/traffic_spmv.txt 2>&1; cut -c1-170 $O/traffic_spmv.txt | head
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
