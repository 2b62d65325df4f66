#!/bin/bash
# projector check on the GPU box: the Radon tests and the forward / adjoint timings over image sizes.  usage: tools/r03_radon.sh <out-subdir>
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
timeout 900 python -m pytest tests/test_gpu_radon_accuracy.py tests/test_gpu_operators.py -q -x -k "radon" 2>&1 | tail -6
python3 tools/radon_micro.py 2>&1 | grep "fwd\|adj"
