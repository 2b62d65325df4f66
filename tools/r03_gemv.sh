#!/bin/bash
# gemv micro-benchmark over the tuning knobs.  usage: tools/r03_gemv.sh <out-subdir>
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
run() { echo "== $*"; env "$@" python3 tools/gemv_micro.py 2>&1 | grep -v amdgpu.ids | grep -v "host thread pools"; }
{
run TRK_GEMVN_UNROLL=4
run TRK_GEMVN_UNROLL=8
run TRK_GEMVN_UNROLL=16
run TRK_GEMVN_UNROLL=8 TRK_GEMVN_GRID=8
run TRK_GEMVN_UNROLL=8 TRK_GEMVN_GRID=2
run TRK_GEMVN_UNROLL=16 TRK_GEMVN_GRID=2
run TRK_GEMVT_PER_CU=4
run TRK_GEMVT_PER_CU=16
run TRK_NT=3
} > $O/gemv_micro.txt 2>&1
cat $O/gemv_micro.txt
