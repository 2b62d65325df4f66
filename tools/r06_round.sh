#!/bin/bash
# Round 6 GPU visit: ONE script, steps by name.  usage: tools/r06_round.sh <out-subdir> [steps...]
# steps: none (build only) | test (whole -m gpu suite, bars logged) | soak (the suite three more times, two of them concurrently) | tradon (projector tests) | radon (4096^2 / 2048^2 / 1024^2 / 512^2 projector rates) |
#        pmc_radon (counters of the 4096^2 pair) | smoke | drv | bench | bench2/4/8 (ranks on one GPU over gloo) | prof | c3 (C3 instrument) | py:<script> [runs tools/<script>] |
#        mb:<name> (builds + runs tools/microbench/<name>.hip) | traffic:<tag>,<script>[,args] | stats:<tag>,<script>[,args] | gaps:<tag>,<n>,<script>[,args]
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1; shift
STEPS=${@:-test}
mkdir -p $O
export TMPDIR=/tmp
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
for s in $STEPS; do case $s in
none) ;;
test)
  rm -f $O/bars.txt
  TRK_BARS_LOG=$O/bars.txt timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log ;;
soak)
  # the whole -m gpu suite three more times: once alone, then two copies side by side on the one GPU (host threads, mailboxes and worker
  # pools under contention) — the round-end gate runs this suite on a fresh box, a flaky test there costs the round
  timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/soak_1.log 2>&1; echo "soak 1 rc=$?"; tail -1 $O/soak_1.log
  (timeout 1800 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/soak_2a.log 2>&1; echo "soak 2a rc=$?") &
  p1=$!
  (timeout 1800 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/soak_2b.log 2>&1; echo "soak 2b rc=$?") &
  p2=$!
  wait $p1 $p2
  tail -1 $O/soak_2a.log; tail -1 $O/soak_2b.log ;;
tradon)
  timeout 1500 python -m pytest tests/test_gpu_radon_accuracy.py tests/test_gpu_operators.py tests/test_gpu_ref64.py -m gpu -x -q > $O/pytest_radon.log 2>&1; echo "pytest radon rc=$?"; tail -6 $O/pytest_radon.log ;;
radon)
  timeout 600 python3 tools/radon_micro.py 4096 2048 1024 512 > $O/radon_micro.txt 2>&1; echo "radon rc=$?"; cat $O/radon_micro.txt ;;
pmc_radon)
  timeout 1500 bash tools/gpu_pmc_cmd.sh k_radon tools/radon_one.py 4096 > $O/radon_4096_pmc.txt 2>&1; echo "pmc rc=$?"; grep -A40 "fwd_quad" $O/radon_4096_pmc.txt | head -60 ;;
smoke)
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log ;;
drv)
  for i in 1 2 3; do
    timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/drv_$i.json 2> $O/drv_$i.err
    python3 -c "
import json
r = json.load(open('$O/drv_$i.json'))
print('drv $i:', r['value'], {k: r['roofline'][k] for k in ('frac', 'avg_kernel_us', 'median_kernel_us', 'min_kernel_us', 'max_kernel_us')})"
  done ;;
bench)
  timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err; echo "bench rc=$?"; cat $O/bench_driver_flags.json ;;
prof)
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_drv -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/prof_drv.json 2> $O/prof_drv.err); echo "prof drv rc=$?"
  f=$(ls -t $O/prof_drv/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/bench_driver_flags_kernel_stats.csv
  head -c 1200 $O/prof_drv.json; echo
  head -8 $O/bench_driver_flags_kernel_stats.csv ;;
bench2|bench4|bench8)
  n=${s#bench}
  TRK_DIST_BACKEND=gloo TRK_SINGLE_DEVICE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus $n --steps 50 --no-cpu-baseline > $O/bench_${n}ranks_one_gpu_gloo.json 2> $O/bench$n.err; echo "bench$n rc=$?"; head -c 2500 $O/bench_${n}ranks_one_gpu_gloo.json; echo; tail -3 $O/bench$n.err ;;
c3)
  timeout 1200 python3 tools/r05_c3_instrument.py 100 > $O/c3_instrument.txt 2> $O/c3_instrument.err; echo "instr rc=$?"; head -40 $O/c3_instrument.txt; tail -3 $O/c3_instrument.err ;;
traffic:*)
  # HBM-side bytes per launch of tools/<script> (separate FETCH_SIZE / WRITE_SIZE passes): traffic:<tag>,<script>[,args]
  n=${s#traffic:}; tag=${n%%,*}; rest=${n#*,}; a=${rest//,/ }
  for ctr in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_${tag}_$ctr -- python3 $R/tools/$a > /dev/null 2>&1); echo "pmc $tag $ctr rc=$?"
  done
  python3 $R/tools/traffic_summary.py $O > $O/traffic_$tag.txt 2>&1; cut -c1-175 $O/traffic_$tag.txt | head -12
  find $O -name "*counter_collection.csv" -delete ;;
stats:*)
  # rocprofv3 --kernel-trace --stats of tools/<script>: stats:<tag>,<script>[,args]
  n=${s#stats:}; tag=${n%%,*}; rest=${n#*,}; a=${rest//,/ }
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -- python3 $R/tools/$a > $O/prof_$tag.log 2>&1); echo "stats $tag rc=$?"
  f=$(ls -t $O/prof_$tag/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/${tag}_kernel_stats.csv; head -12 $O/${tag}_kernel_stats.csv | cut -c1-160 ;;
gaps:*)
  # busy time / gaps of the last <n> launches of tools/<script> (rocprofv3 --kernel-trace + tools/trace_gaps.py): gaps:<tag>,<n>,<script>[,args]
  n=${s#gaps:}; tag=${n%%,*}; rest=${n#*,}; cnt=${rest%%,*}; rest=${rest#*,}; a=${rest//,/ }
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/gaps_$tag -- python3 $R/tools/$a > $O/gaps_$tag.log 2>&1); echo "gaps $tag rc=$?"
  python3 $R/tools/trace_gaps.py $(ls -t $O/gaps_$tag/*/*kernel_trace.csv | head -1) $cnt > $O/gaps_$tag.txt 2>&1; cat $O/gaps_$tag.txt ;;
py:*)
  n=${s#py:}; a=${n//,/ }; f=${a%% *}
  timeout 1500 python3 tools/$a > $O/${f%.py}.txt 2>&1; echo "$f rc=$?"; tail -40 $O/${f%.py}.txt ;;
mb:*)
  n=${s#mb:}
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -Itrips_py_amd/csrc -o /tmp/$n tools/microbench/$n.hip > $O/$n.build.log 2>&1 && timeout 900 /tmp/$n > $O/$n.txt 2>&1; echo "$n rc=$?"; tail -60 $O/$n.txt ;;
esac; done
find $O -name "*kernel_trace.csv" -delete 2>/dev/null
find $O -name "*.db" -delete 2>/dev/null
du -sh $O | tail -1
