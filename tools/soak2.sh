#!/bin/bash
# Second soak (GPU box, at most 19 minutes): the sweeps added in round 4 -- every from-scratch fit split against the oracle, CSG rebuilds, parity under
# the other reduction order -- plus the plain parity sweep on fresh seeds.  usage: bash tools/soak2.sh <tag> [first seed]   -> gpurun_out/<tag>/fuzz_*.txt
TAG=${1:-soak2}; S0=${2:-1000}; O=gpurun_out/$TAG; mkdir -p $O
(timeout -k 10 400 python tools/fuzz_split.py 400 $S0 > $O/fuzz_split.txt 2>&1; tail -1 $O/fuzz_split.txt)
(timeout -k 10 200 python tools/fuzz_csg.py 200 $S0 > $O/fuzz_csg.txt 2>&1; tail -1 $O/fuzz_csg.txt)
(HPSDF_REDUCTION_ORDER=left timeout -k 10 240 python tools/fuzz_parity.py 400 $((S0 + 30000)) > $O/fuzz_parity_left.txt 2>&1; tail -1 $O/fuzz_parity_left.txt)
(timeout -k 10 240 python tools/fuzz_parity.py 400 $((S0 + 40000)) > $O/fuzz_parity.txt 2>&1; tail -1 $O/fuzz_parity.txt)
