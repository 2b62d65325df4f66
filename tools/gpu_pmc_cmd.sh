#!/bin/bash
# usage: tools/gpu_pmc_cmd.sh <kernel-name-substring> <python script> [args]  -> mean PMC counters per dispatch of matching kernels
R=$GRAFT_REPO_ROOT
F=$1; shift
O=$R/gpurun_out/pmc_cmd
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
            "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_WAVES" \
            "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SMEM" \
            "TCP_PENDING_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_TCC_READ_REQ_sum" \
            "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $O/p$i -- python3 $R/$1 ${@:2} > $O/p$i.log 2>&1
  echo "pass $i rc=$?"
done
python3 - <<PY
import csv, glob, os, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "$F" in k:
            acc[k[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"    {c:32s} mean {sum(v)/len(v):18.1f}  (n={len(v)})")
PY
