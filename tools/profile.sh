#!/bin/bash
# Run on the GPU box from the repo root: kernel-trace stats of bench.py, then two PMC passes (HBM read / write
# bytes of query_kernel).  Outputs under gpurun_out/prof_<tag>/ ; tools/summarize_profile.py condenses them.
set -u
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write  # (a merged gpurun_out/ may hold an earlier call's files: the summary must read this call's)
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
# the trace pass runs bench.py exactly as the driver does (default flags): the kernel averages of the summary are those of the bench line
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_pmc_write.log 2>&1
cd $REPO && python3 tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
tail -40 $OUT/summary.txt
