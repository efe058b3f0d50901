mkdir -p gpurun_out/r03z
(timeout -k 10 900 python tools/fuzz_mesh_bvh.py 4000 100000 > gpurun_out/r03z/fuzz_mesh_bvh.txt 2>&1; tail -2 gpurun_out/r03z/fuzz_mesh_bvh.txt)
(timeout -k 10 900 python tools/fuzz_parity.py 1200 3000 > gpurun_out/r03z/fuzz_parity.txt 2>&1; tail -2 gpurun_out/r03z/fuzz_parity.txt)
(timeout -k 10 600 python tools/fuzz_continuity.py 60 > gpurun_out/r03z/fuzz_continuity.txt 2>&1; tail -2 gpurun_out/r03z/fuzz_continuity.txt)
(timeout -k 10 600 python tools/fuzz_mesh_parity.py > gpurun_out/r03z/fuzz_mesh_parity.txt 2>&1; tail -2 gpurun_out/r03z/fuzz_mesh_parity.txt)
