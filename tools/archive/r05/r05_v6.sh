#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/v6; mkdir -p $O; export TMPDIR=/tmp; cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?"
for nt in 0 1; do for cap in 0 16 64; do for g in 0 -1 -2; do
  echo "== TRK_CSR_NT=$nt TRK_CSR_GRID_PER_CU=$cap TRK_CSR_GROUP=$g"
  TRK_CSR_NT=$nt TRK_CSR_GRID_PER_CU=$cap TRK_CSR_GROUP=$g timeout 300 python3 tools/spmv_micro.py 2>/dev/null | cut -c1-260
done; done; done | tee $O/spmv_sweep.txt
timeout 300 python3 tools/gmres_rates.py 2>/dev/null | tee $O/gmres_rates.txt
