#!/bin/bash
# GPU box, from the repo root: kernel timeline of Create through the device-side frontier.
set -u
TAG=${1:-r02}
OUT=$PWD/gpurun_out/frontier_$TAG
mkdir -p $OUT
REPO=${GRAFT_REPO_ROOT:-/root/repo}
python3 $REPO/tools/frontier_trace.py > $OUT/plain.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/tools/frontier_trace.py > $OUT/traced.log 2>&1
cd $REPO
f=$(ls $OUT/trace/*/*kernel_trace.csv | head -1)
python3 - "$f" > $OUT/timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last Create of the first config: find the last fr_init_kernel before the 13th
inits = [i for i, r in enumerate(rows) if "fr_init_kernel" in r["Kernel_Name"]]
for which, label in ((11, "union3 @ 1e-5, 12th Create"), (23, "union3 @ 1e-7, 12th Create")):
    if which >= len(inits): continue
    a = inits[which]; b = inits[which + 1] if which + 1 < len(inits) else len(rows)
    t0 = int(rows[a]["Start_Timestamp"])
    print(label)
    prev_end = t0
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].replace("hpsdf::", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
        print("  +%8.1f us  gap %6.1f  dur %7.1f us  grid %-8s %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r.get("Grid_Size", "?"), name))
        prev_end = e
    print("  span %.1f us" % ((prev_end - t0) / 1e3))
PY
cat $OUT/plain.log | tail -30; cat $OUT/timeline.txt | head -80
