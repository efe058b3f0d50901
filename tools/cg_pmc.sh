#!/bin/bash
# SQ / TA / TCP counters of the continuity solve's kernels (tools/continuity_probe.py).  Usage: bash tools/cg_pmc.sh <tag>
TAG=${1:-cg}
OUT=$PWD/gpurun_out/pmccg_$TAG
mkdir -p $OUT
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY" \
         "SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM" \
         "TA_TA_BUSY_sum TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
         "TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -- python3 $REPO/tools/continuity_probe.py 2 > $OUT/log$i.txt 2>&1
done
cd $REPO && python3 - "$OUT" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
acc = {}
for f in glob.glob(os.path.join(out, "**/*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "cg_" not in r["Kernel_Name"]:
            continue
        acc.setdefault((r["Kernel_Name"][:40], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    if "step1" in k[0] or "spmv_aux" in k[0] or "step2" in k[0]:
        print("%-42s %-30s n=%d avg %.4g max %.4g" % (k[0], k[1], len(v), sum(v) / len(v), max(v)))
PY
