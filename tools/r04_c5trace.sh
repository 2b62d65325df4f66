#!/bin/bash
# C5 (GKS and CGLS on 32 frames of 256^2 x 15 angles, one rank) behind rocprofv3 --kernel-trace: the last launches by kernel, busy fraction, gaps.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/c5trace
export TMPDIR=/tmp
mkdir -p $O
cd /tmp
for t in c5_gks_trace c5_cgls_trace; do
  rm -rf $O/p
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/p -- python3 $R/tools/$t.py > $O/$t.log 2>&1
  f=$(ls -t $O/p/*/*kernel_trace.csv | head -1)
  echo "=== $t"; python3 $R/tools/trace_gaps.py $f 300 | cut -c1-120
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
