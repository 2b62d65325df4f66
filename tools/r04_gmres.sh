#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/gmres; export TMPDIR=/tmp; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_solvers.py tests/test_gpu_history.py tests/test_gpu_edges.py tests/test_gpu_core_abi.py -m gpu -x -q 2>&1 | tail -3
python3 $R/tools/gmres_rates.py 512 2>/dev/null
python3 $R/tools/gmres_rates.py 2048 2>/dev/null
cd /tmp; rm -rf $O/p; timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/p -- python3 $R/tools/gmres_trace.py 512 > $O/t.log 2>&1
python3 $R/tools/trace_gaps.py $(ls -t $O/p/*/*kernel_trace.csv | head -1) 120 | tail -3
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
