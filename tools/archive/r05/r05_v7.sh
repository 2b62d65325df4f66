#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/v7; mkdir -p $O; export TMPDIR=/tmp; cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
TRK_BARS_LOG=$O/bars.txt timeout 1500 python -m pytest tests/test_gpu_solvers.py tests/test_gpu_operators.py tests/test_gpu_history.py tests/test_gpu_random_shapes.py tests/test_gpu_radon_accuracy.py tests/test_gpu_mmgks_fullsize.py tests/test_gpu_ref64.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 300 python3 tools/gmres_rates.py 2>/dev/null | tee $O/gmres_rates.txt
timeout 300 python3 tools/spmv_micro.py 2>/dev/null | cut -c1-250 | tee $O/spmv_micro.txt
for g in 8 4; do echo "TRK_CSR_GROUP=$g (long rows)"; TRK_CSR_GROUP=$g timeout 300 python3 tools/spmv_micro.py 2>/dev/null | cut -c1-250 | head -2; done | tee -a $O/spmv_micro.txt
