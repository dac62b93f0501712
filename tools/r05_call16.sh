#!/bin/bash
# round 5: generated models through per-node rows / classes / the in-kernel loop; then the whole GPU suite
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_ode_file_gpu.py -x -q -m gpu > gpurun_out/r05_tests16a.log 2>&1; rc=$?; echo "ode_file rc $rc"; tail -30 gpurun_out/r05_tests16a.log
[ $rc = 0 ] && timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r05_full7.log 2>&1; echo "full rc $?"; tail -5 gpurun_out/r05_full7.log
