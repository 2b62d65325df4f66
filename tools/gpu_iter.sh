#!/bin/bash
# quick GPU iteration: parity tests + short bench (+ optional kernel trace)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 100 --warmup 10 ${BENCH_ARGS} > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?"
cat gpurun_out/bench.log; tail -5 gpurun_out/bench.err
if [ -n "$PROFILE" ]; then
  R=$GRAFT_REPO_ROOT
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bench --output-format csv -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $R/gpurun_out/prof_bench.log 2>&1; echo "rocprof rc=$?"
fi
