set -u
O=gpurun_out/r03z; mkdir -p $O
timeout -k 10 700 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
timeout -k 10 500 bash tools/profile.sh r03z > $O/profile_log.txt 2>&1; tail -5 $O/profile_log.txt
timeout -k 10 200 python bench.py > $O/bench.json 2> $O/bench_err.txt; cut -c1-600 $O/bench.json
