#!/bin/bash
# round 5: generated kernels of a big synthetic model (spilled registers): tests, then its kernel time
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_ode_file_gpu.py -x -q -m gpu > gpurun_out/r05_tests25.log 2>&1; echo "ode_file rc $?"; tail -12 gpurun_out/r05_tests25.log
python3 tools/bench_from_ode.py tests/data/big_cell.ode --n 4194304 2>&1 | tee gpurun_out/r05_from_ode_big.txt
