#!/bin/bash
# round 5: the open decomposed solve against the waiting one (three mailbox ranks), then the whole GPU suite
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_distributed_gpu.py -x -q -m gpu -k "open_decomposed" > gpurun_out/r05_tests22.log 2>&1; rc=$?; echo "open-dist test rc $rc"; tail -5 gpurun_out/r05_tests22.log
[ $rc = 0 ] || exit 1
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r05_full9.log 2>&1; echo "full rc $?"; tail -5 gpurun_out/r05_full9.log
