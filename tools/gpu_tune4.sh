#!/bin/bash
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
for rpb in 28 46 64 82; do
  echo "== RPB=$rpb"; TRK_BLUR_RPB=$rpb python3 $R/tools/blur_micro.py 4096 30 2>&1 | grep -E "^blur"
done
cd /tmp; timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc4 -- python3 $R/tools/blur_micro.py 4096 5 > /dev/null 2>&1
cd $R; python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc4/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_blur_slide" in r["Kernel_Name"]: acc[r["Kernel_Name"][:60]+r["Counter_Name"]].append(float(r["Counter_Value"]))
print({k: round(sum(v)/len(v)) for k, v in acc.items()})
PY
timeout 600 python -m pytest tests/test_gpu_blur.py tests/test_gpu_fullsize.py -m gpu -q 2>&1 | tail -3
