#!/bin/bash
# End-of-round soak on the GPU box (at most 19 minutes: the four time limits add up to less than one gpurun call): the four randomised sweeps, each against the oracle or the O(n) scan.
# usage: bash tools/soak.sh <tag> [first seed of fuzz_mesh_bvh] [first seed of fuzz_parity]   -> gpurun_out/<tag>/fuzz_*.txt
TAG=${1:-soak}; BVH0=${2:-100000}; PAR0=${3:-3000}; O=gpurun_out/$TAG; mkdir -p $O
(timeout -k 10 420 python tools/fuzz_mesh_bvh.py 4000 $BVH0 > $O/fuzz_mesh_bvh.txt 2>&1; tail -2 $O/fuzz_mesh_bvh.txt)
(timeout -k 10 330 python tools/fuzz_parity.py 1200 $PAR0 > $O/fuzz_parity.txt 2>&1; tail -2 $O/fuzz_parity.txt)
(timeout -k 10 180 python tools/fuzz_continuity.py 60 > $O/fuzz_continuity.txt 2>&1; tail -2 $O/fuzz_continuity.txt)
(timeout -k 10 200 python tools/fuzz_mesh_parity.py > $O/fuzz_mesh_parity.txt 2>&1; tail -2 $O/fuzz_mesh_parity.txt)
