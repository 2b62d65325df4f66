#!/bin/bash
# usage: tools/gpu_prof_cmd.sh <python script + args>   -> per-kernel stats of that run (rocprofv3 --kernel-trace --stats)
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
rm -rf $R/gpurun_out/prof_cmd
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cmd -- python3 $R/$1 ${@:2} > $R/gpurun_out/prof_cmd.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$R/gpurun_out/prof_cmd/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:20]:
        print(f"{r['Name'][:84]:84s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.2f} min={float(r['MinNs'])/1e3:8.1f} max={float(r['MaxNs'])/1e3:8.1f} tot_ms={float(r['TotalDurationNs'])/1e6:8.2f}")
PY
