#!/bin/bash
# Build the C++ examples against include/ + libhpsdf.so with plain g++ and run them (GPU box, repo root):
#   bash tools/cxx_examples.sh <outdir>     -> <outdir>/cxx_benchmarks.txt, cxx_unit_tests.txt, cxx_meshing_benchmarks.txt
OUT=${1:-gpurun_out/cxx}; mkdir -p $OUT
LIBDIR=$PWD/hp-adaptive-signed-distance-field-octree_amd/lib
for n in hp_benchmarks hp_unit_tests meshing_benchmarks; do
  g++ -std=c++17 -O2 -Wall -Wno-comment -I include examples/$n.cpp -o /tmp/$n -L $LIBDIR -lhpsdf -Wl,-rpath,$LIBDIR -Wl,-rpath,/opt/rocm/lib -L/opt/rocm/lib -lamdhip64 -pthread || exit 1
done
timeout -k 10 200 /tmp/hp_benchmarks > $OUT/cxx_benchmarks.txt 2>&1; echo "hp_benchmarks rc=$?"
timeout -k 10 300 /tmp/hp_unit_tests > $OUT/cxx_unit_tests.txt 2>&1; echo "hp_unit_tests rc=$?"
TMPDIR=/tmp timeout -k 10 300 /tmp/meshing_benchmarks > $OUT/cxx_meshing_benchmarks.txt 2>&1; echo "meshing_benchmarks rc=$?"
