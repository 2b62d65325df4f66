#!/bin/bash
# The Infinity-Cache experiment of VERDICT r03 item 4b: ordinary / non-temporal stores in the producers (TRK_NT mask: bit 0 blur
# output, bit 1 x', bit 2 r, bit 3 p; 192 = the basis-row load hints, irrelevant here) x forward / reverse sweep of the two update
# kernels (TRK_REV).  Headline loop only, driver flags; rates and the per-kernel averages of one profiled run each.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1; mkdir -p $O
cd $R
for cfg in "default:" "rev:TRK_REV=1" "plain:TRK_NT=192" "plain_rev:TRK_NT=192 TRK_REV=1" "ntall:TRK_NT=207" "ntall_rev:TRK_NT=207 TRK_REV=1" "blur_only:TRK_NT=193" "blur_only_rev:TRK_NT=193 TRK_REV=1"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  for i in 1 2; do
    env $envs timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/mall_${name}_$i.json 2> $O/mall_${name}_$i.err
  done
  python3 - <<PY
import json
v = [json.load(open("$O/mall_${name}_%d.json" % i)) for i in (1, 2)]
print("%-14s %-26s iterations/s %s   forward-blur kernel us %s" % ("$name", "$envs", [r["value"] for r in v], [r["roofline"]["avg_kernel_us"] for r in v]))
PY
done
