#!/bin/bash
# VALU-issue counters of the fit micro-benchmark in its default mode, union3 field, one degree after the other -> fit_pmc.json
# (bench.py's fit_microbench.roofline).  Usage: bash tools/fit_pmc_all.sh <tag> [degrees...]
TAG=${1:-fit}; shift
DEGS=${@:-2 3 4 5 6 7 8}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUTALL=$PWD/gpurun_out/pmcf_$TAG
mkdir -p $OUTALL
ARGS=""
for D in $DEGS; do
  CELLS=16384; [ $D -le 3 ] && CELLS=65536
  OUT=$OUTALL/p$D
  mkdir -p $OUT
  ( cd /tmp && export TMPDIR=/tmp
    i=0
    for C in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" \
             "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU"; do
      i=$((i+1))
      rocprofv3 --pmc $C --output-format csv -d $OUT/c$i -- python3 $REPO/tools/fit_one.py union3 $D $CELLS default > $OUT/log$i.txt 2>&1
    done )
  ARGS="$ARGS $D=$OUT"
  tail -1 $OUT/log1.txt
done
cd $REPO && python3 tools/pmc_json.py fit "$OUTALL/fit_pmc.json" "$TAG" $ARGS > $OUTALL/summary.txt && echo "wrote $OUTALL/fit_pmc.json (copy to profiles/fit_pmc.json)"
python3 - "$OUTALL/fit_pmc.json" <<'PY'
import json, sys
r = json.load(open(sys.argv[1]))
for p, d in r["degrees"].items():
    print(p, {k: round(v["frac_valu_issue"], 3) if v.get("frac_valu_issue") else None for k, v in d.items()})
PY
