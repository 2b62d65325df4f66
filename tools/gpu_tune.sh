#!/bin/bash
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
for rpb in 16 32 64 128 256; do
  echo "== TRK_BLUR_RPB=$rpb"; TRK_BLUR_RPB=$rpb python3 $R/tools/blur_micro.py 4096 30 2>&1 | grep -E "^blur"
done
mkdir -p $R/gpurun_out/pmc2; cd /tmp
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" \
            "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/gpurun_out/pmc2/p$i -- python3 $R/tools/blur_micro.py 4096 5 > $R/gpurun_out/pmc2/p$i.log 2>&1
  echo "pass $i rc=$?"
done
cd $R; python3 tools/pmc_summary.py gpurun_out/pmc2 > gpurun_out/pmc2/summary.txt 2>&1
grep -A26 "k_blur_slide<9, 9, 6, true>" gpurun_out/pmc2/summary.txt | head -30
