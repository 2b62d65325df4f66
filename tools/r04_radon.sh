#!/bin/bash
# projector check on the GPU box (round 4): the Radon tests, the kernel variants against each other, forward / adjoint timings.
# usage: tools/r04_radon.sh <out-subdir> [sizes...]
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1; shift
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
timeout 900 python -m pytest tests/test_gpu_radon_accuracy.py tests/test_gpu_operators.py -q -x -k "radon" 2>&1 | tail -6
timeout 300 python3 tools/radon_variants_check.py 2>&1 | tail -14
python3 tools/radon_micro.py ${@:-512 1024 2048 4096} 2>&1 | tee $O/radon_micro.txt | grep "fwd\|adj"
TRK_RADON_NO_QUAD=1 python3 tools/radon_micro.py 4096 2>&1 | grep "fwd" | sed 's/^/NO_QUAD /'
