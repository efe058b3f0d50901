#!/bin/bash
# The evidence of a build, as committed under profiles/ at the end of a round.  Run on the GPU box from the repo root in two calls
# (each fits one gpurun call):  bash tools/round_evidence.sh <tag> 1   -- GPU tests, tools/profile.sh (trace + FETCH / WRITE passes
# of the default bench command), the bench line;  ... <tag> 2 -- mesh probe, mesh sampler counters, the reference's benchmark list,
# frontier check, device-vs-host mesh preparation, traversal statistics of a diagnostic build (LAST: it replaces the library in
# this copy of the tree);  ... <tag> 3 -- fit counters, Create's timeline, the query kernels' floors, ordered point sets, default Config().  Outputs: gpurun_out/<tag>/ and gpurun_out/prof_<tag>/; copy what is to be judged into profiles/.
set -u
TAG=${1:-evidence}; PART=${2:-1}
O=gpurun_out/$TAG; mkdir -p $O
if [ "$PART" = 1 ]; then
  timeout -k 10 700 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
  timeout -k 10 500 bash tools/profile.sh $TAG > $O/profile_log.txt 2>&1; tail -5 $O/profile_log.txt
  timeout -k 10 200 python bench.py > $O/bench.json 2> $O/bench_err.txt; cut -c1-600 $O/bench.json
elif [ "$PART" = 3 ]; then
  # counters of the fits (-> fit_pmc.json, fit_mfma_pmc.json), Create's kernel timeline, the any-tree query kernels link by link (lab library,
  # built beforehand with build.py --lab), Query on ordered point sets, the default Config()
  timeout -k 10 420 bash tools/fit_pmc_all.sh $TAG > $O/fit_pmc.txt 2>&1; tail -8 $O/fit_pmc.txt
  timeout -k 10 200 bash tools/fit_mfma_pmc.sh $TAG > $O/fit_mfma_pmc.txt 2>&1; tail -4 $O/fit_mfma_pmc.txt
  timeout -k 10 120 bash tools/frontier_trace.sh $TAG > $O/frontier_trace.txt 2>&1; tail -30 $O/frontier_trace.txt
  timeout -k 10 150 python tools/query_general_floor.py > $O/query_general_floor.txt 2>&1; cat $O/query_general_floor.txt
  timeout -k 10 100 python tools/query_patterns.py > $O/query_patterns.txt 2>&1; cat $O/query_patterns.txt
  timeout -k 10 100 python tools/default_config_create.py 0 > $O/default_config.txt 2>&1; cut -c1-200 $O/default_config.txt
else
  MESH_PROBE_TARGETS=1e-6,1e-7 timeout -k 10 300 python tools/mesh_probe.py 5 7 8 torus > $O/mesh_probe.txt 2>&1; tail -12 $O/mesh_probe.txt
  timeout -k 10 200 bash tools/mesh_pmc.sh ${TAG}_8 8 > $O/mesh_pmc_8.txt 2>&1; tail -6 $O/mesh_pmc_8.txt
  timeout -k 10 200 bash tools/mesh_pmc.sh ${TAG}_torus torus > $O/mesh_pmc_torus.txt 2>&1; tail -6 $O/mesh_pmc_torus.txt
  timeout -k 10 120 python tools/ref_benchmarks.py > $O/ref_benchmarks.txt 2>&1; cat $O/ref_benchmarks.txt
  timeout -k 10 200 python tools/frontier_check.py > $O/frontier_check.txt 2>&1; tail -16 $O/frontier_check.txt
  timeout -k 10 120 python tools/mesh_prepare_probe.py > $O/mesh_prepare.txt 2>&1; tail -6 $O/mesh_prepare.txt
  HPSDF_EXTRA_FLAGS="-DHPSDF_MESH_STATS_BUILD -DHPSDF_MESH_POOL_STATS" timeout -k 10 400 python hp-adaptive-signed-distance-field-octree_amd/build.py --force > $O/stats_build.txt 2>&1; tail -2 $O/stats_build.txt
  HPSDF_MESH_STATS=1 timeout -k 10 200 python tools/mesh_probe.py 8 torus > $O/mesh_traversal_stats.txt 2>&1; cat $O/mesh_traversal_stats.txt
fi
