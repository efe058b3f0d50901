#!/bin/bash
# A/B of two builds on one box: lib/libhpsdf.so (new) against lib/libhpsdf_prev.so, alternating.  usage: tools/ab_builds.sh <python script> [args]
L=hp-adaptive-signed-distance-field-octree_amd/lib
cp $L/libhpsdf.so $L/libhpsdf_new.so
for round in 1 2; do
  for v in new prev; do
    cp $L/libhpsdf_$v.so $L/libhpsdf.so
    echo "== $v (round $round)"
    python3 "$@" || exit 1
  done
done
cp $L/libhpsdf_new.so $L/libhpsdf.so
