#!/bin/bash
# Round 3, first GPU visit: why does `bench.py --steps 20 --warmup 5` (the driver's flags) time slower forward launches than
# the 100-step runs?  usage: tools/r03_diag1.sh <out-subdir>
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
mkdir -p $O
export TMPDIR=/tmp
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc
for i in 1 2 3 4 5; do
  timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/drv_$i.json 2> $O/drv_$i.err
  python3 - <<EOF
import json
r = json.load(open("$O/drv_$i.json"))
print("drv $i:", r["value"], {k: r["roofline"][k] for k in ("frac", "avg_kernel_us", "min_kernel_us") if k in r["roofline"]}, {k: v for k, v in r["roofline"].items() if "median" in k or "p90" in k or "run_in" in k})
EOF
done
timeout 300 python3 tools/bench_ramp.py 3200 0 > $O/ramp_cold.txt 2>&1; cat $O/ramp_cold.txt
timeout 300 python3 tools/bench_ramp.py 800 3000 > $O/ramp_idle3s.txt 2>&1; cat $O/ramp_idle3s.txt
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/drv_full.json 2> $O/drv_full.err; echo "full rc=$?"; head -c 1500 $O/drv_full.json; echo
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
