#!/bin/bash
# PMC passes over the blur micro-driver (counters in their own runs, kernel-trace only)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
export TMPDIR=/tmp
cd /tmp
python3 $R/tools/blur_micro.py 4096 20 > $R/gpurun_out/pmc/micro.log 2>&1
cat $R/gpurun_out/pmc/micro.log | tail -5
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
            "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_WAVES" \
            "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/gpurun_out/pmc/p$i -- python3 $R/tools/blur_micro.py 4096 5 > $R/gpurun_out/pmc/p$i.log 2>&1
  echo "pass $i rc=$?"
done
cd $R
python3 tools/pmc_summary.py gpurun_out/pmc > gpurun_out/pmc/summary.txt 2>&1
cat gpurun_out/pmc/summary.txt | head -60
