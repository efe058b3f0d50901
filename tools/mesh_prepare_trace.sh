#!/bin/bash
# GPU box: per-kernel times of the device mesh preparation (tools/mesh_prepare_probe.py)
OUT=$PWD/gpurun_out/meshprep_${1:-r02}
mkdir -p $OUT
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/tools/mesh_prepare_probe.py > $OUT/log.txt 2>&1
cd $REPO
f=$(ls $OUT/trace/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print("%-70s calls %4s  avg %10.1f us  total %8.2f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
tail -4 $OUT/log.txt
