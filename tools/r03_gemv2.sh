#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
run() { echo "== $*"; env "$@" python3 tools/gemv_micro.py 2>&1 | grep -v amdgpu.ids | grep -v "host thread pools"; }
{
run TRK_X=0
run TRK_GEMVN_UNROLL=16
run TRK_GEMVN_UNROLL=16 TRK_GEMVN_GRID=4
} > $O/gemv_micro.txt 2>&1
cat $O/gemv_micro.txt
