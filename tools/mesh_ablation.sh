#!/bin/bash
# What mesh_sample_kernel's phases cost (lab builds, -DHPSDF_MESH_ABL=n: one phase left out or run twice; the values of those builds
# are NOT the field's).  Two steps: (1) HERE, before gpurun (hipcc cross-compiles):  bash tools/mesh_ablation.sh build
# (2) on the GPU box: bash tools/mesh_ablation.sh run > gpurun_out/mesh_ablation.txt
set -u
cd "$(dirname "$0")/.."
VARIANTS="abl1:-DHPSDF_MESH_ABL=1 abl2:-DHPSDF_MESH_ABL=2 abl5:-DHPSDF_MESH_ABL=5 abl6:-DHPSDF_MESH_ABL=6 abl9:-DHPSDF_MESH_ABL=9 nowalk:-DHPSDF_SEED_WALK=0 noexch:-DHPSDF_SEED_EXCHANGE=0"
if [ "${1:-}" = build ]; then
    python3 hp-adaptive-signed-distance-field-octree_amd/build.py > /dev/null || exit 1
    pids=""
    for v in $VARIANTS; do
        python3 hp-adaptive-signed-distance-field-octree_amd/build.py --variant=$v > /dev/null &
        pids="$pids $!"
        if [ $(echo $pids | wc -w) -ge 4 ]; then wait $pids; pids=""; fi
    done
    wait $pids
    ls -la hp-adaptive-signed-distance-field-octree_amd/lib/
    exit 0
fi
echo "== as it is"
MESH_PROBE_TARGETS=1e-6 python3 tools/mesh_probe.py torus 2>&1 | grep -v amdgpu.ids
for v in $VARIANTS; do
    n=${v%%:*}
    echo "== $v"
    HPSDF_LIBRARY=$n MESH_PROBE_TARGETS=1e-6 python3 tools/mesh_probe.py torus 2>&1 | grep -v amdgpu.ids
done
