#!/bin/bash
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_one -- python3 $R/tools/configs_micro.py $1 > $R/gpurun_out/prof_one.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$R/gpurun_out/prof_one/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:16]:
        print(f"{r['Name'][:84]:84s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.2f} min={float(r['MinNs'])/1e3:8.1f} max={float(r['MaxNs'])/1e3:8.1f} tot_ms={float(r['TotalDurationNs'])/1e6:8.2f} pct={r['Percentage']}")
PY
