#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
mkdir -p $O
export TMPDIR=/tmp
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
{
python3 tools/c4_rate.py
TRK_GEMVN_UNROLL=4 TRK_GEMVN_GRID=0 python3 tools/c4_rate.py
TRK_GEMVT_PER_CU=8 python3 tools/c4_rate.py
TRK_GEMVN_UNROLL=16 python3 tools/c4_rate.py
} 2>&1 | grep "MMGKS it/s" | tee $O/c4_rates.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4 -- python3 $R/tools/c4_rate.py > $O/prof_c4.log 2>&1); echo "prof rc=$?"
f=$(ls -t $O/prof_c4/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/c4_kernel_stats.csv
cut -c1-170 $O/c4_kernel_stats.csv | head -16
find $O -name "*kernel_trace.csv" -delete 2>/dev/null; find $O -name "*.db" -delete 2>/dev/null
