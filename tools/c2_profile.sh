cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_c2 -- python3 $GRAFT_REPO_ROOT/tools/configs_micro.py c2 > $GRAFT_REPO_ROOT/gpurun_out/prof_c2.log 2>&1; cd $GRAFT_REPO_ROOT; python3 - <<EOP
import csv, glob, os
f = max(glob.glob("gpurun_out/prof_c2/*/*kernel_stats.csv"), key=os.path.getmtime)
for r in list(csv.DictReader(open(f)))[:4]:
    print(r["Name"][:80], r["Calls"], round(float(r["AverageNs"])/1e3,2), round(float(r["MinNs"])/1e3,2))
EOP
grep -- "->" gpurun_out/prof_c2.log
