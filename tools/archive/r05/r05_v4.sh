#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/v4; mkdir -p $O; export TMPDIR=/tmp; cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
TRK_BARS_LOG=$O/bars.txt timeout 900 python -m pytest tests/test_gpu_operators.py tests/test_gpu_kernels.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
grep sparse $O/bars.txt
timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err; echo "bench rc=$?"
python3 - <<PY
import json
r=json.load(open("$O/bench_driver_flags.json"))
print("value", r["value"], "roofline", r["roofline"]["frac"])
for k,v in r["extra"].items():
    if isinstance(v, dict):
        print(k, {kk: vv for kk, vv in v.items() if kk in ("iters_per_sec_all_ranks","fwd_us","adj_us","error","gks_iters_per_sec_all_ranks","cgls_iters_per_sec_all_ranks")})
print(json.dumps(r["extra"].get("next_sparse_dynamic"), indent=1)[:3000])
print(json.dumps(r["extra"]["c4_mmgks_tv_4096"]["roofline"], indent=1)[:1500])
PY
TRK_DIST_BACKEND=gloo TRK_SINGLE_DEVICE=1 timeout 900 python bench.py --gpus 8 --steps 50 --no-cpu-baseline > $O/bench_8ranks_one_gpu_gloo.json 2> $O/bench8.err; echo "bench8 rc=$?"; head -c 1500 $O/bench_8ranks_one_gpu_gloo.json; echo; tail -3 $O/bench8.err
