set -u
O=gpurun_out/r03z; mkdir -p $O
MESH_PROBE_TARGETS=1e-6,1e-7 timeout -k 10 300 python tools/mesh_probe.py 5 7 8 torus > $O/mesh_probe.txt 2>&1; tail -12 $O/mesh_probe.txt
timeout -k 10 200 bash tools/mesh_pmc.sh r03z_8 8 > $O/mesh_pmc_8.txt 2>&1; tail -6 $O/mesh_pmc_8.txt
timeout -k 10 200 bash tools/mesh_pmc.sh r03z_torus torus > $O/mesh_pmc_torus.txt 2>&1; tail -6 $O/mesh_pmc_torus.txt
timeout -k 10 120 python tools/ref_benchmarks.py > $O/ref_benchmarks.txt 2>&1; cat $O/ref_benchmarks.txt
timeout -k 10 200 python tools/frontier_check.py > $O/frontier_check.txt 2>&1; tail -16 $O/frontier_check.txt
timeout -k 10 120 python tools/mesh_prepare_probe.py > $O/mesh_prepare.txt 2>&1; tail -6 $O/mesh_prepare.txt
HPSDF_EXTRA_FLAGS="-DHPSDF_MESH_STATS_BUILD -DHPSDF_MESH_POOL_STATS" timeout -k 10 400 python hp-adaptive-signed-distance-field-octree_amd/build.py --force > $O/stats_build.txt 2>&1; tail -2 $O/stats_build.txt
HPSDF_MESH_STATS=1 timeout -k 10 200 python tools/mesh_probe.py 8 torus > $O/mesh_traversal_stats.txt 2>&1; cat $O/mesh_traversal_stats.txt
