#!/bin/bash
# first GPU contact: smoke, parity tests, a short bench, a kernel-trace profile
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" 
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -30 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 50 --warmup 5 > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?"
cat gpurun_out/bench.log; tail -5 gpurun_out/bench.err
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_bench --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT; find gpurun_out/prof_bench -name "*stats*" | head; 
