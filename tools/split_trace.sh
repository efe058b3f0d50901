#!/bin/bash
# per-kernel durations of the split fit (exact top-degree rows + matrix-core lower rows): rocprofv3 kernel trace of the micro-benchmark
# usage (GPU box, repo root): bash tools/split_trace.sh <tag> "<field> <degree> <cells> <mode>" ...
TAG=$1; shift
OUT=$PWD/gpurun_out/trace_$TAG; mkdir -p $OUT
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for ARGS in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t$i -- python3 $REPO/tools/fit_one.py $ARGS > $OUT/log$i.txt 2>&1
  echo "== $ARGS: $(grep TFLOP $OUT/log$i.txt)"
  python3 - $OUT/t$i <<'PY'
import csv, glob, os, sys
for f in glob.glob(os.path.join(sys.argv[1], "**/*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "fit" in r["Name"]:
            print("   %-100s calls %3s avg %9.1f us" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
