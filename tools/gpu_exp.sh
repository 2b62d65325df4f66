#!/bin/bash
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc3
echo "--- normal"; python3 $R/tools/blur_micro.py 4096 30 | grep blur
cd /tmp
for c in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
 timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc3/n_${c%% *} -- python3 $R/tools/blur_micro.py 4096 5 > /dev/null 2>&1
done
cp $R/trips_py_amd/csrc/libtrk.so /tmp/libtrk_keep.so
cp $R/gpurun_out/libtrk_nolr.so $R/trips_py_amd/csrc/libtrk.so
touch $R/trips_py_amd/csrc/libtrk.so
echo "--- no L/R loads (timing experiment, wrong numerics)"; python3 $R/tools/blur_micro.py 4096 30 | grep blur
for c in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
 timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc3/x_${c%% *} -- python3 $R/tools/blur_micro.py 4096 5 > /dev/null 2>&1
done
cp /tmp/libtrk_keep.so $R/trips_py_amd/csrc/libtrk.so
cd $R
python3 - <<'PY'
import csv, glob, collections
for tag in ("n", "x"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/pmc3/{tag}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_blur_slide<9, 9, 6, false>" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(tag, {k: round(sum(v)/len(v)) for k, v in acc.items()})
PY
