#!/bin/bash
# Matrix-core counters of the opt-in fast fit (fit_mfma_kernel, hpsdf_ctx_set_fit_mode(HPSDF_FIT_FAST)): SQ_VALU_MFMA_BUSY_CYCLES against
# the kernel's active cycles, per degree, with the headline field (union3) and with a field that costs nothing (plane: contraction only)
# -> fit_mfma_pmc.json (bench.py's fit_microbench.*.fast_fit.mfma_busy).  Usage: bash tools/fit_mfma_pmc.sh <tag> [degrees...]
TAG=${1:-mfma}; shift
DEGS=${@:-4 6 8}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUTALL=$PWD/gpurun_out/pmcx_$TAG
mkdir -p $OUTALL
ARGS=""
for D in $DEGS; do
  for F in union3 plane; do
    OUT=$OUTALL/p${D}_$F
    mkdir -p $OUT
    ( cd /tmp && export TMPDIR=/tmp
      i=0
      for C in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVES" "SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS"; do
        i=$((i+1))
        rocprofv3 --pmc $C --output-format csv -d $OUT/c$i -- python3 $REPO/tools/fit_one.py $F $D 16384 fast > $OUT/log$i.txt 2>&1
      done )
    ARGS="$ARGS $D:$F=$OUT"
    tail -1 $OUT/log1.txt
  done
done
cd $REPO && python3 tools/pmc_json.py mfma "$OUTALL/fit_mfma_pmc.json" "$TAG" $ARGS > $OUTALL/summary.txt && echo "wrote $OUTALL/fit_mfma_pmc.json (copy to profiles/fit_mfma_pmc.json)"
python3 - "$OUTALL/fit_mfma_pmc.json" <<'PY'
import json, sys
r = json.load(open(sys.argv[1]))
for p, d in r["degrees"].items():
    print(p, {f: {k: (round(v, 3) if isinstance(v, float) else v) for k, v in x.items() if k in ("mfma_busy", "frac_valu_issue")} for f, x in d.items()})
PY
