#!/bin/bash
# round 5: the per-node tile kernel with its forward coefficient rows in the tile layout (aligned 512-byte pieces per wave and plane):
# the per-node tests, then the shell A/B (BEAT_VTL_TILED = 0 | 1)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_var_gpu.py -x -q -m gpu > gpurun_out/r05_tests44.log 2>&1; rc=$?; echo "var tests rc $rc"; tail -4 gpurun_out/r05_tests44.log
[ $rc = 0 ] || exit 1
for rep in 1 2 3; do for v in 0 1; do
  echo -n "BEAT_VTL_TILED=$v  "; BEAT_VTL_TILED=$v timeout -k 10 300 python tools/bench_biv.py --size 400 --steps 20 2>&1 | tail -1
done; done | tee gpurun_out/r05_biv400_tiled.txt
