#!/bin/bash
# round 5, fifth GPU call: the right-hand side of the per-node path on the tiles (two single-window passes over B and A rows)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_var_gpu.py tests/test_guess_gpu.py -x -q -m gpu > gpurun_out/r05_tests5.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r05_tests5.log
for v in 1 0 1 0; do
  echo "BEAT_VTL_RHS=$v"; BEAT_VTL_RHS=$v timeout -k 10 300 python tools/bench_biv.py --size 400 --steps 20 2>&1 | tail -1
done | tee gpurun_out/r05_biv400_rhs.txt
BEAT_VTL_RHS=1 timeout -k 10 200 python tools/bench_voxel.py --n 400 2>&1 | tail -6 | tee gpurun_out/r05_voxel400.txt
BEAT_VTL_RHS=0 timeout -k 10 200 python tools/bench_voxel.py --n 400 2>&1 | tail -6 | tee -a gpurun_out/r05_voxel400.txt
