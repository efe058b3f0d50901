#!/bin/bash
# PMC passes over the fit micro-benchmark.  Usage: bash tools/fit_pmc.sh <tag> <field> <degree> <cells> [fast]
TAG=${1:-fit}; FIELD=${2:-plane}; DEG=${3:-2}; CELLS=${4:-65536}; MODE=${5:-default}
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY" \
         "SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" \
         "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_FLAT SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -- python3 $REPO/tools/fit_one.py $FIELD $DEG $CELLS $MODE > $OUT/log$i.txt 2>&1
done
cd $REPO && python3 - "$OUT" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
acc = {}
for f in glob.glob(os.path.join(out, "**/*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "fit_kernel" not in r["Kernel_Name"] and "fit_mfma_kernel" not in r["Kernel_Name"]:
            continue
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print("%-34s n=%d avg %.4g" % (k, len(v), sum(v) / len(v)))
a = lambda k: sum(acc[k]) / len(acc[k]) if k in acc else None
if a("SQ_VALU_MFMA_BUSY_CYCLES") and a("SQ_BUSY_CYCLES"):
    # SQ_BUSY_CYCLES counts per shader engine (x 32 on this chip), MFMA busy cycles per SIMD: see MI355X_MICROARCH.md, PMC notes
    print("MFMA busy / (GRBM_GUI_ACTIVE x 1024 SIMDs) = %.3f" % (a("SQ_VALU_MFMA_BUSY_CYCLES") / (a("GRBM_GUI_ACTIVE") / 8 * 1024)) if a("GRBM_GUI_ACTIVE") else "")
PY
cat $OUT/log1.txt | tail -2
