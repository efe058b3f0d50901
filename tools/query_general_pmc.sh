#!/bin/bash
# SQ / TA / TCP counters of query_general_kernel on union3 @ 1e-7.  Usage: bash tools/query_general_pmc.sh <tag>
TAG=${1:-g}
OUT=$PWD/gpurun_out/pmcg_$TAG
mkdir -p $OUT
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY" \
         "SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" \
         "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_FLAT" \
         "TA_TA_BUSY_sum TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -- python3 $REPO/tools/query_one.py 1e-7 > $OUT/log$i.txt 2>&1
done
cd $REPO && python3 - "$OUT" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
acc = {}
for f in glob.glob(os.path.join(out, "**/*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "query_general" not in r["Kernel_Name"] and "query_deep" not in r["Kernel_Name"] and "defer_scan" not in r["Kernel_Name"]:
            continue
        acc.setdefault(r["Kernel_Name"][:36] + " " + r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print("%-34s n=%d avg %.4g" % (k, len(v), sum(v) / len(v)))
PY
