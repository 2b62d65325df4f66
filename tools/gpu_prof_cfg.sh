#!/bin/bash
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
for c in c4 c5 c3; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$c -- python3 $R/tools/configs_micro.py $c > $R/gpurun_out/prof_$c.log 2>&1
  echo "== $c"; tail -8 $R/gpurun_out/prof_$c.log | grep -v amdgpu
  python3 - <<PY
import csv, glob
for f in glob.glob("$R/gpurun_out/prof_$c/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:14]:
        print(f"{r['Name'][:90]:90s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.2f} tot_ms={float(r['TotalDurationNs'])/1e6:9.2f} pct={r['Percentage']}")
PY
done
